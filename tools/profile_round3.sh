#!/bin/bash
# Evidence run for profiles/r03: the driver's bench line (bf16 default with extras, roofline + cpu_baseline), rocprofv3 kernel-trace
# stats + timeline of the graph-replayed step, the step's longest dependency chain, PMC traffic passes, per-layer conv tables.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 500 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json || { tail -5 $O/bench_default.err; exit 1; }
echo "default done"; cut -c1-300 $O/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/prof_graph.log 2>&1 || exit 1
f=$(ls $O/prof_graph/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f 80 > $O/graph_timeline.txt; python3 tools/timeline_tail.py $f 1200 > $O/graph_timeline_tail.txt
cp $(ls $O/prof_graph/*/*kernel_stats.csv | head -1) $O/graph_kernel_stats.csv
rm -rf $O/prof_graph
echo "trace done"; head -3 $O/graph_timeline.txt
timeout -k 10 200 python tools/graph_dot.py $O/graph.dot > $O/graph_dot.log 2>&1 || { tail -5 $O/graph_dot.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --mangled-kernels --output-format csv -d $O/prof_cp -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/prof_cp.log 2>&1 || exit 1
f=$(ls $O/prof_cp/*/*kernel_trace.csv | head -1)
python3 tools/graph_critical_path.py $O/graph.dot $f --nodes > $O/critical_path_nodes.txt 2>&1
python3 tools/graph_critical_path.py $O/graph.dot $f > $O/critical_path.txt 2>&1
rm -rf $O/prof_cp $O/graph.dot
echo "chain done"; grep "longest chain" $O/critical_path.txt
timeout -k 5 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extras > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 5 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extras > $O/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write ${GIT_HEAD:-unknown} > $O/pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
echo "pmc done"; head -c 400 $O/pmc_traffic.json
L=b1c2,b2c2,b3c2,b4c2,b5c2,b1c1,b2c1
timeout -k 10 120 python tools/conv_bench.py $L 30 2>&1 | grep -v amdgpu.ids > $O/conv_layers_fwd.txt
WGRAD_REP=8 timeout -k 10 120 python tools/conv_bench.py $L 30 wgrad 2>&1 | grep -v amdgpu.ids > $O/conv_layers_wgrad.txt
cat $O/conv_layers_fwd.txt $O/conv_layers_wgrad.txt
