cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_write.log 2>&1
