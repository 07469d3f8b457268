#!/bin/bash
# round 3, run i: the step's longest dependency chain on the current build (graph dot + mangled kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 200 python tools/graph_dot.py $O/graph_now.dot > $O/graph_dot.log 2>&1 || { tail -5 $O/graph_dot.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --mangled-kernels --output-format csv -d $O/prof_cp -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/prof_cp.log 2>&1 || { tail -5 $O/prof_cp.log; exit 1; }
f=$(ls $O/prof_cp/*/*kernel_trace.csv | head -1)
python3 tools/graph_critical_path.py $O/graph_now.dot $f > $O/critical_path.txt 2>&1
rm -rf $O/prof_cp
head -60 $O/critical_path.txt
