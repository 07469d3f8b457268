#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
b() { echo -n "$* : "; env AST_FUSED_FINALIZE=0 AST_OLD_TOTAL=1 AST_OLD_ADAM=1 AST_ARENA_RESET=0 "$@" timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms; }
{
for i in 1 2; do
  echo -n "r2tree : "; (cd _ab_r2 && timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  b A=1
  b AST_NO_SYNC_HYPER=1
  b AST_NO_SYNC_HYPER=1 AST_NO_TOKCHECK=1
done
echo -n "now launch (all toggles): "; AST_FUSED_FINALIZE=0 AST_OLD_TOTAL=1 AST_OLD_ADAM=1 AST_ARENA_RESET=0 AST_NO_SYNC_HYPER=1 AST_NO_TOKCHECK=1 python tools/launch_cost.py 2>/dev/null | tail -1
} | tee $O/bisect4.txt
