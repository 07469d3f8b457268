#!/bin/bash
# round 3, run h: deferred weight gradients -- one flush stream for all banks or one per bank; then the whole GPU suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b AST_WGRAD_DEFER=0; b AST_WGRAD_DEFER=1; b AST_WGRAD_DEFER=1 AST_PARALLEL_FLUSH=0; b AST_WGRAD_DEFER=1 AST_WGRAD_SLABS=0; done; } | tee $O/ab_defer_flush.txt
timeout -k 10 700 python -m pytest tests -m gpu -q -rA --maxfail=8 > $O/t10.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/t10.txt | tail -12
