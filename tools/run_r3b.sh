#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad or conv" > $O/t6a.txt 2>&1; tail -3 $O/t6a.txt
if grep -q "Memory access fault" $O/t6a.txt; then echo FAULT; exit 1; fi
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms; }
{ for i in 1 2 3; do b AST_WGRAD_SLABS=0; b AST_WGRAD_SLABS=128; b AST_WGRAD_SLABS=128 AST_WGRAD_WG_TARGET=512; done; } | tee $O/ab_slabs.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -rA --maxfail=8 > $O/t6.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/t6.txt | tail -12
