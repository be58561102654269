cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d /root/repo/gpurun_out/pmc_conv -- python /root/repo/tools/conv_bench.py 4 256 > /root/repo/gpurun_out/pmc_conv.log 2>&1 &&
rocprofv3 --kernel-trace --pmc $C --output-format csv -d /root/repo/gpurun_out/pmc_gemm -- python /root/repo/tools/kernel_bench.py --only "gemm fwd" --rounds 1 > /root/repo/gpurun_out/pmc_gemm.log 2>&1
