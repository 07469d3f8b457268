#!/bin/bash
# round 3, run d: rows kernel at its best workgroup target, and the convolution weight gradients on one side stream
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b AST_WGRAD_ROWS=0; b AST_WGRAD_ROWS=1; b AST_WGRAD_ROWS=1 AST_WGRAD_STREAM=1; b AST_WGRAD_ROWS=0 AST_WGRAD_STREAM=1; b AST_WGRAD_ROWS=1 AST_WGRAD_STREAM=1 AST_WGRAD_SLABS=0; done; } | tee $O/ab_wstream.txt
