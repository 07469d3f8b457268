#!/bin/bash
# round 3, run j: existing knobs re-judged with the weight gradients off the data-gradient chain
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b A=0; b AST_FUSED_FINALIZE=1; b AST_TOK_PROGRAMS=1; b AST_TOK_PROGRAMS=2; b AST_WGRAD_REPLICAS=4; b AST_WGRAD_WG_TARGET=256; done; } | tee $O/ab_knobs_deferred.txt
