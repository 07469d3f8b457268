#!/bin/bash
# round 3, final: full GPU suite (-rA teed), smoke(), then the evidence pack of tools/profile_round3.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -rA > $O/gpu_tests_final.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests_final.txt | tail -8
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/profile_round3.sh
timeout -k 10 200 python tools/layer_profile.py 2>&1 | grep -v amdgpu.ids > $O/gemm_layers.txt; head -3 $O/gemm_layers.txt
