#!/bin/bash
# round 3, run f: timeline of the step with the deferred weight gradients (tail) and without
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
prof() {  # name, env...
  local name=$1; shift
  export "$@"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/prof_$name.log 2>&1 || return 1
  cp $(ls $O/prof_$name/*/*kernel_stats.csv | head -1) $O/kstats_$name.csv
  f=$(ls $O/prof_$name/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f 80 > $O/timeline_$name.txt; python3 tools/timeline_tail.py $f 2500 > $O/tail_$name.txt
  rm -rf $O/prof_$name
}
prof defer1 AST_WGRAD_DEFER=1 && prof defer0 AST_WGRAD_DEFER=0
head -3 $O/timeline_defer1.txt $O/timeline_defer0.txt
