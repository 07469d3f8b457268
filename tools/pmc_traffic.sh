cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 5 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch.log 2>&1 && \
timeout -k 5 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/pmc_write.log 2>&1
ls -la gpurun_out/pmc_fetch/*/ | head; du -sh gpurun_out/pmc_fetch gpurun_out/pmc_write
