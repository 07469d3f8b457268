#!/bin/bash
# round 3, run o: number of hardware queues the HIP runtime maps the streams onto
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -3 $O/err.txt; }
{ for i in 1 2; do b A=0; b GPU_MAX_HW_QUEUES=3; b GPU_MAX_HW_QUEUES=5; b GPU_MAX_HW_QUEUES=6; b GPU_MAX_HW_QUEUES=1; done; } | tee $O/ab_hw_queues.txt
