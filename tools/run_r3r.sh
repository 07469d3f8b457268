#!/bin/bash
# round 3, run r: patch kernel with all taps' weights resident per channel slab (WALL): parity, per-layer and whole-step A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv or pconv or igemm" > $O/t14a.txt 2>&1; tail -3 $O/t14a.txt
if grep -q "Memory access fault" $O/t14a.txt; then echo FAULT; exit 1; fi
if grep -q "failed" $O/t14a.txt; then grep -E "^E |^FAILED" $O/t14a.txt | head -20; exit 1; fi
L=b1c2,b2c2,b3c2,b4c2,b5c2
{ for w in 0 2 auto; do env $( [ $w = auto ] && echo A=1 || echo AST_PCONV_WALL=$w ) timeout -k 10 120 python tools/conv_bench.py $L 30 2>&1 | grep -v amdgpu.ids | sed "s/^/wall=$w /"; done; } | tee $O/pconv_wall_layers2.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b AST_PCONV_WALL=0; b A=auto; done; } | tee $O/ab_wall2.txt
