#!/bin/bash
# A/B timing of bench.py FLAG sets on ONE box: tools/ab_flags.sh "--no-cqt" "" ...  (each argument is one flag string)
out=gpurun_out/ab_flags.txt; : > $out
for rep in 1 2; do
for f in "$@"; do
  v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline $f 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])") || exit 1
  echo "[$f] -> $v ms" | tee -a $out
done
done
