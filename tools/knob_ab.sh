#!/bin/bash
# In-graph A/B of a tuning knob by KERNEL time: tools/knob_ab.sh 'kernel-name regex' "ENV=a" "ENV=b" ...
# Each environment runs bench.py under rocprofv3 --kernel-trace --stats; prints calls / total ns / average ns of the matching
# kernels in the replayed steps (co-running kernels, cold caches: what isolated back-to-back launches do not show).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
re=$1; shift
out=gpurun_out/knob_ab.txt; : > $out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/knob$i -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/knob$i.log 2>&1 || exit 1
  f=$(ls gpurun_out/knob$i/*/*kernel_stats.csv | head -1)
  echo "== $e" >> $out
  grep -E "$re" $f | python3 -c "
import sys, csv
tot = 0; n = 0
for r in csv.reader(sys.stdin):
    print(f'  {r[0][:70]:70s} calls {r[1]:>5s} avg {float(r[3])/1e3:7.2f} us'); tot += int(r[2]); n += int(r[1])
print(f'  total {tot/1e3:.0f} us over {n} calls')" >> $out
  rm -rf gpurun_out/knob$i
done
cat $out
