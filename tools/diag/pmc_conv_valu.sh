R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_conv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_conv/valu -- python3 $R/tools/conv_bench.py 4 256 0,1,2 > $R/gpurun_out/pmc_conv_valu.log 2>&1
