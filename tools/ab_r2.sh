#!/bin/bash
# Same-box A/B of the whole step: the round-2 tree (_ab_r2/, built from commit 9c8fc2b) against the current tree with the statistics
# finalize folded into the apply passes (AST_FUSED_FINALIZE=1) and without; then rocprofv3 kernel statistics of all three.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "r2tree "; (cd _ab_r2 && timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "now fusedfin=0 "; AST_FUSED_FINALIZE=0 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms
  echo -n "now fusedfin=1 "; AST_FUSED_FINALIZE=1 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms
done | tee $O/ab_r2.txt
prof() {  # name, dir, env
  local name=$1 dir=$2; shift 2
  (cd $dir && env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$name -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline $EXTRA > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1) || return 1
  cp $(ls $O/prof_$name/*/*kernel_stats.csv | head -1) $O/kstats_$name.csv
  f=$(ls $O/prof_$name/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f 80 > $O/timeline_$name.txt
  rm -rf $O/prof_$name
}
EXTRA="" prof r2tree _ab_r2 A=1 && EXTRA="--no-extras" prof fin0 . AST_FUSED_FINALIZE=0 && EXTRA="--no-extras" prof fin1 . AST_FUSED_FINALIZE=1
echo prof done; ls -la $O/kstats_*.csv
