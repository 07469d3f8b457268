#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
b() { echo -n "$* : "; env AST_FUSED_FINALIZE=0 "$@" timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms; }
{
echo -n "r2tree launch: "; (cd _ab_r2 && python tools/launch_cost.py 2>/dev/null | tail -1)
echo -n "now launch: "; AST_FUSED_FINALIZE=0 python tools/launch_cost.py 2>/dev/null | tail -1
for i in 1 2; do
  echo -n "r2tree : "; (cd _ab_r2 && timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  b A=1
  b AST_OLD_TOTAL=1
  b AST_OLD_ADAM=1
  b AST_ARENA_RESET=0
  b AST_OLD_TOTAL=1 AST_OLD_ADAM=1 AST_ARENA_RESET=0
  b AST_FUSED_BN_STATS=0
done
} | tee $O/bisect.txt
