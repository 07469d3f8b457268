cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_graph
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_graph.log 2>&1 || exit 1
python tools/timeline.py gpurun_out/prof_graph/*/*_kernel_trace.csv > gpurun_out/tl4.txt 2>&1
