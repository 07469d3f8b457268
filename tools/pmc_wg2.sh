cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=$1; K=${2:-wgrad}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR TCC_EA0_ATOMIC_sum"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcw_${L}_$tag -- python3 tools/one_layer.py $L 3 wgrad > gpurun_out/pmcw_${L}_$tag.log 2>&1 || exit 1
  python3 tools/pmc_summ.py gpurun_out/pmcw_${L}_$tag $K
done
