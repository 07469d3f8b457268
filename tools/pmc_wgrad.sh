#!/bin/bash
# PMC passes over the isolated weight-gradient launch of one layer: HBM-side fetch, L2 hits/misses, LDS conflicts
# usage: tools/pmc_wgrad.sh <layer> <tag>   (AST_WGRAD_ROWS etc. from the environment)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=${1:-b1c2}; T=${2:-rows}; O=gpurun_out/r3/pmc_$T; mkdir -p $O
export WGRAD_SLABS=128
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$O/$(echo $pass | tr ' ' '_')
  timeout -k 5 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python3 tools/conv_bench.py $L 5 wgrad > $d.log 2>&1 || { echo "pass $pass failed"; tail -3 $d.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("$L $T", k[0], k[1], "mean", sum(v) / len(v), "n", len(v))
PY
