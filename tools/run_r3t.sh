#!/bin/bash
# round 3, run t: deferred weight gradients balanced by lending the larger banks' cheapest launches to the smallest bank's stream
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x -k "scheduling" > $O/t16a.txt 2>&1; tail -2 $O/t16a.txt
grep -q " passed" $O/t16a.txt || { grep -E "^E " $O/t16a.txt | head; exit 1; }
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b A=0; b AST_WGRAD_DEFER_POOL=1 AST_WGRAD_DEFER_LEND=0.5; b AST_WGRAD_DEFER_POOL=1 AST_WGRAD_DEFER_LEND=1.0; b AST_WGRAD_DEFER_POOL=1 AST_WGRAD_DEFER_LEND=1.5; done; } | tee $O/ab_lend.txt
