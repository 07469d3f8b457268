#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -rA --maxfail=8 > $O/t5.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/t5.txt | tail -12
if grep -q "Memory access fault" $O/t5.txt; then echo FAULT; exit 1; fi
bash tools/rehearse_dp.sh
timeout -k 10 200 python tools/dp_match.py --batch 8 --sections 1 --matched 1 2>/dev/null | tail -1 | tee $O/dp_match_matched.json | cut -c1-700
timeout -k 10 300 python bench.py > $O/b3.json 2> $O/b3.err; tail -c 300 $O/b3.json; tail -2 $O/b3.err
