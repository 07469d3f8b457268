#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
R=$GRAFT_REPO_ROOT
{
for i in 1 2; do
  echo -n "r2 python + r2 lib : "; (cd _ab_r2 && timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "r2 python + c1 lib : "; (cd _ab_r2 && AST_HIP_LIB=$R/_ab_c1/audio-style-transfer_amd/ast_amd/libast_hip.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "r2 python + now lib : "; (cd _ab_r2 && AST_HIP_LIB=$R/audio-style-transfer_amd/ast_amd/libast_hip.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "c1 python + c1 lib : "; (cd _ab_c1 && timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms)
done
} | tee $O/bisect3.txt
