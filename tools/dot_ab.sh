#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
(cd _ab_r2 && python tools/graph_dot.py $GRAFT_REPO_ROOT/$O/graph_r2.dot > /dev/null 2>&1)
AST_FUSED_FINALIZE=0 python tools/graph_dot.py $O/graph_now.dot > /dev/null 2>&1
ls -la $O/*.dot
python tools/dot_summary.py $O/graph_r2.dot
python tools/dot_summary.py $O/graph_now.dot
head -c 3000 $O/graph_now.dot
