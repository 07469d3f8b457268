#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms; }
{
for i in 1 2 3; do
  echo -n "r2tree : "; (cd _ab_r2 && timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  b A=1
  b AST_COLLAPSE_JOINS=1
done
} | tee $O/ab_collapse.txt
timeout -k 10 800 python -m pytest tests -m gpu -q -rA --maxfail=8 > $O/t4.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/t4.txt | tail -12
