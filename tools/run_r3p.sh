#!/bin/bash
# round 3, run p: reconstruction loss with slotted partial sums + finishing launch; scheduling-modes test
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "loss or recon or comprehensive or model or decoder" > $O/t12a.txt 2>&1; tail -3 $O/t12a.txt
timeout -k 10 400 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x -k "scheduling or modes_agree" > $O/t12b.txt 2>&1; tail -3 $O/t12b.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'], d['losses']['rec'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3 4; do b A=0; done; } | tee $O/ab_recon.txt
