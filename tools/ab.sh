#!/bin/bash
# A/B timing on ONE box: tools/ab.sh "ENV=a ENV2=b" "ENV=c" ...  (each argument is an environment for one bench run)
out=gpurun_out/ab.txt; : > $out
for rep in 1 2; do
for e in "$@"; do
  v=$(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])") || exit 1
  echo "$e -> $v ms" | tee -a $out
done
done
