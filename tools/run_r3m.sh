#!/bin/bash
# round 3, run m: pooled deferred launches with the hand-over through the origin stream
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
AST_WGRAD_DEFER_POOL=1 timeout -k 10 300 python -m pytest tests/test_gpu_bench_config.py -m gpu -q -x > $O/t11p.txt 2>&1; tail -2 $O/t11p.txt
grep -q " passed" $O/t11p.txt || exit 1
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b A=0; b AST_WGRAD_DEFER_POOL=1; done; } | tee $O/ab_pool.txt
