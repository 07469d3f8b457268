#!/bin/bash
# round 3, run n: HIP runtime switches that touch graph replay and queue mapping (environment only), and the capture reproducer's new pattern
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
PATTERNS=sibling,sibling_mutual timeout -k 10 200 python tools/capture_forks.py --all 4 16 > $O/forks_mutual.txt 2>&1; cat $O/forks_mutual.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2; do b A=0; b GPU_MAX_HW_QUEUES=8; b GPU_MAX_HW_QUEUES=2; b HIP_FORCE_DEV_KERNARG=0; b HIP_FORCE_DEV_KERNARG=1; b DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; b DEBUG_HIP_GRAPH_NUM_STREAMS=8; done; } | tee $O/ab_runtime_env.txt
