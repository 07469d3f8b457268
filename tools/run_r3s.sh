#!/bin/bash
# round 3, run s: 4-fragment patch tiles for the 5x10 images of the last block (WALL), parity + per-layer + whole step against the previous commit's tree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x > $O/t15a.txt 2>&1; tail -3 $O/t15a.txt
if grep -q "Memory access fault" $O/t15a.txt; then echo FAULT; exit 1; fi
if grep -q "failed" $O/t15a.txt; then grep -E "^E |^FAILED" $O/t15a.txt | head -20; exit 1; fi
timeout -k 10 120 python tools/conv_bench.py b3c2,b4c2,b5c2 30 2>&1 | grep -v amdgpu.ids | tee $O/pconv_tiny.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
{ for i in 1 2 3 4; do
  echo -n "prev : "; (cd _ab_prev && timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "now  : "; timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms
done; } | tee $O/ab_prev2.txt
