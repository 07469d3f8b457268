#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
for i in 1 2; do
  echo -n "r2tree : "; (cd _ab_r2 && timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "c1 : "; (cd _ab_c1 && timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "c2 tap=0 : "; (cd _ab_c2 && AST_WGRAD_TAP=0 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "now fin0 : "; AST_FUSED_FINALIZE=0 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms
  echo -n "now fin1 : "; AST_FUSED_FINALIZE=1 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms
done
} | tee $O/bisect2.txt
