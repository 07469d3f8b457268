#!/bin/bash
# round 3, run g: deferred weight gradients on 1 / 2 / 3 streams per bank
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
AST_WGRAD_DEFER=1 AST_WGRAD_DEFER_STREAMS=3 timeout -k 10 300 python -m pytest tests/test_gpu_bench_config.py -m gpu -q -x > $O/t9b.txt 2>&1; tail -3 $O/t9b.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b AST_WGRAD_DEFER=0; for k in 1 2 3 4; do b AST_WGRAD_DEFER=1 AST_WGRAD_DEFER_STREAMS=$k; done; done; } | tee $O/ab_defer_streams.txt
