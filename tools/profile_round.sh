#!/bin/bash
# Evidence run for profiles/: default bench line, rocprofv3 kernel-trace stats of the graph-replayed step, inference bench.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py 2>gpurun_out/bench_full_err.txt | tail -1 > gpurun_out/bench_full.json || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_graph.log 2>&1 || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --infer 2>/dev/null | tail -1 > gpurun_out/bench_infer.json || exit 1
ls gpurun_out/prof_graph/*/
