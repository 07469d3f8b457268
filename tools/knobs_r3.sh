#!/bin/bash
# Re-check the step-structure knobs on the current build (same box, alternating): which of round 2's scheduling choices still pay.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline $FLAGS 2>/dev/null | ms; }
{
for i in 1 2; do
  FLAGS="" b A=1
  FLAGS="" b AST_PARALLEL_FLUSH=0
  FLAGS="" b AST_EARLY_PREPARE=0
  FLAGS="" b AST_BRANCH_ORDER=syc
  FLAGS="" b AST_BRANCH_ORDER=csy
  FLAGS="" b AST_FRONTEND_FORK=1
  FLAGS="--no-overlap-d" b A=1
  FLAGS="" b AST_WGRAD_REPLICAS=1
  FLAGS="" b AST_FUSED_BN_BWD=0
done
} | tee $O/knobs.txt
