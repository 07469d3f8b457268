#!/bin/bash
# round 3, run k: ring depth of the rows kernel and the pooled distribution of the deferred launches, judged in the step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
for st in 2 3; do AST_WGRAD_ROWS_STAGES=$st timeout -k 10 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad_rows" > $O/t11_$st.txt 2>&1; tail -1 $O/t11_$st.txt; if grep -q "Memory access fault" $O/t11_$st.txt; then echo FAULT; exit 1; fi; done
AST_WGRAD_DEFER_POOL=1 timeout -k 10 300 python -m pytest tests/test_gpu_bench_config.py -m gpu -q -x > $O/t11p.txt 2>&1; tail -2 $O/t11p.txt
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['losses']['total'])"; }
b() { echo -n "$* : "; env "$@" timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>$O/err.txt | ms || tail -5 $O/err.txt; }
{ for i in 1 2 3; do b A=0; b AST_WGRAD_ROWS_STAGES=3; b AST_WGRAD_ROWS_STAGES=2; b AST_WGRAD_DEFER_POOL=1; b AST_WGRAD_DEFER_POOL=1 AST_WGRAD_ROWS_STAGES=3; done; } | tee $O/ab_stages_pool.txt
