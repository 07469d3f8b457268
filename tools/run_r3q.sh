#!/bin/bash
# round 3, run q: same-box A/B against the tree of the previous commit (_ab_prev/): reconstruction-loss finishing launch + bilinear backward
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bilinear" > $O/t13a.txt 2>&1; tail -2 $O/t13a.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
{ for i in 1 2 3 4; do
  echo -n "prev : "; (cd _ab_prev && timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms)
  echo -n "now  : "; timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | ms
done; } | tee $O/ab_prev.txt
