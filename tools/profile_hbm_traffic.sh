# HBM bytes per launch per kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py
# (MI355X_MICROARCH.md, HBM section) -> gpurun_out/pmc_fetch, gpurun_out/pmc_write; run on the GPU box via gpurun
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_write.log 2>&1
