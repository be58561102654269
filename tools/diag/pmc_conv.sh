# SQ / LDS / L1 counters of the GridNet-width convolutions (tools/conv_bench.py shapes 0-2), one rocprofv3 --pmc pass per
# counter group -> gpurun_out/pmc_conv/<first counter>/.   gpurun -- bash tools/diag/pmc_conv.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_conv/$n -- python3 $R/tools/conv_bench.py 4 256 0,1,2 > $R/gpurun_out/pmc_conv_$n.log 2>&1 || exit 1
done
