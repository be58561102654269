cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d /root/repo/gpurun_out/pmc_x3 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --dtype f32x3 > /root/repo/gpurun_out/pmc_x3.log 2>&1
