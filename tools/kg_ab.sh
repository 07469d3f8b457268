cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 4 8; do
  AST_IGEMM_KG_KCH=$k timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kg$k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/kg$k.log 2>&1 || exit 1
  f=$(ls gpurun_out/kg$k/*/*kernel_stats.csv | head -1); grep "igemm_kernelIDF16bLi64ELi64ELi2ELi2ELi[48]ELi2ELi4" $f | cut -c1-200 > gpurun_out/kg$k.txt; tail -1 gpurun_out/kg$k.log | cut -c1-200 >> gpurun_out/kg$k.txt
  rm -rf gpurun_out/kg$k
done
cat gpurun_out/kg4.txt gpurun_out/kg8.txt
