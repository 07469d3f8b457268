#!/bin/bash
# Evidence run for profiles/r02: bench lines (default bf16 with roofline + cpu_baseline, f32 parity mode, inference S=2 and S=4,
# mixed-length stream), rocprofv3 kernel-trace stats of the graph-replayed step, PMC traffic passes.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 400 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json || exit 1
echo "default done"; cut -c1-200 $O/bench_default.json
timeout -k 10 200 python bench.py --dtype f32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f32.json || exit 1
timeout -k 10 300 python bench.py --infer --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_infer_s2.json || exit 1
timeout -k 10 300 python bench.py --infer --sections 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_infer_s4.json || exit 1
timeout -k 10 300 python bench.py --mixed --steps 28 --warmup 7 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_mixed.json || exit 1
echo "benches done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_graph.log 2>&1 || exit 1
f=$(ls $O/prof_graph/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f 80 > $O/graph_timeline.txt
cp $(ls $O/prof_graph/*/*kernel_stats.csv | head -1) $O/graph_kernel_stats.csv
echo "trace done"
timeout -k 5 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 5 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write ${GIT_HEAD:-unknown} > $O/pmc_traffic.json
echo "pmc done"; head -c 600 $O/pmc_traffic.json
rm -rf $O/prof_graph/*/*kernel_trace.csv $O/pmc_fetch $O/pmc_write      # the raw traces are large; the summaries stay
