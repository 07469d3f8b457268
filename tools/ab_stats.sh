L=b0ds,b0c1,b0c2,b1c2,dec0,dec1
OLD=$PWD/audio-style-transfer_amd/ast_amd/libast_hip_old.so
O=gpurun_out/stats_ab.txt; : > $O
echo "== old, no stats" >> $O; NOSTATS=1 AST_HIP_LIB=$OLD timeout -k 10 120 python tools/conv_bench.py $L 30 >> $O 2>&1 || exit 1
echo "== old, BN slot table" >> $O; AST_HIP_LIB=$OLD timeout -k 10 120 python tools/conv_bench.py $L 30 >> $O 2>&1 || exit 1
echo "== new, BN slot table" >> $O; timeout -k 10 120 python tools/conv_bench.py $L 30 >> $O 2>&1 || exit 1
echo "== old, per image" >> $O; PERIMG=1 AST_HIP_LIB=$OLD timeout -k 10 120 python tools/conv_bench.py b0ds,b1c2 30 >> $O 2>&1 || exit 1
echo "== new, per image" >> $O; PERIMG=1 timeout -k 10 120 python tools/conv_bench.py b0ds,b1c2 30 >> $O 2>&1 || exit 1
