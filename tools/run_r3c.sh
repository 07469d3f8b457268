#!/bin/bash
# round 3, run c: the line-staged (LDS-DMA ring) weight-gradient kernel of the 3x3 stride-1 layers -- parity first, then per-layer and whole-step A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad_rows" > $O/t8a.txt 2>&1; tail -5 $O/t8a.txt
if grep -q "Memory access fault" $O/t8a.txt; then echo FAULT; exit 1; fi
if ! grep -q " passed" $O/t8a.txt || grep -q "failed" $O/t8a.txt; then echo ROWS-PARITY-FAILED; grep -E "^E " $O/t8a.txt | head -20; exit 1; fi
L=b1c2,b2c2,b3c2,b4c2,b5c2
{ for ring in 0 1; do for sl in 0 128; do
    AST_WGRAD_ROWS=$ring WGRAD_SLABS=$sl WGRAD_REP=8 timeout -k 10 120 python tools/conv_bench.py $L 30 wgrad 2>&1 | grep -v amdgpu.ids | sed "s/^/rows=$ring slabs=$sl /"
  done; done
  for tg in 256 384 768 1024; do
    AST_WGRAD_ROWS=1 AST_WGRAD_WG_TARGET=$tg WGRAD_SLABS=128 timeout -k 10 120 python tools/conv_bench.py $L 30 wgrad 2>&1 | grep -v amdgpu.ids | sed "s/^/rows=1 slabs=128 target=$tg /"
  done; } | tee $O/wg_rows_layers.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms; }
{ for i in 1 2 3; do b AST_WGRAD_ROWS=0; b AST_WGRAD_ROWS=1; b AST_WGRAD_ROWS=1 AST_WGRAD_SLABS=0; done; } | tee $O/ab_rows.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -rA --maxfail=8 > $O/t8.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/t8.txt | tail -12
