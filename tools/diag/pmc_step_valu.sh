# vector-ALU vs MFMA instruction counts per kernel of the layout-token step (one pipe on gfx950: 8 cycles per vector instruction
# against 64 per v_mfma_f32_32x32x2_f32) -> gpurun_out/pmc_step/.   gpurun -- bash tools/diag/pmc_step_valu.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_step
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_step/valu -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras > $R/gpurun_out/pmc_step_valu.log 2>&1
