# HBM bytes per launch per kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py
# (MI355X_MICROARCH.md, HBM section) -> gpurun_out/pmc_fetch, gpurun_out/pmc_write; then tools/pmc_traffic.py writes
# profiles/<tag>_pmc_hbm_traffic.csv + profiles/pmc_traffic.json.   gpurun -- bash tools/profile_hbm_traffic.sh r02_a
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras > $O/pmc_write.log 2>&1 &&
cd $R && python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $TAG > $O/pmc_traffic_$TAG.log 2>&1 && mkdir -p $O/profiles_out && cp profiles/${TAG}_pmc_hbm_traffic.csv profiles/pmc_traffic.json $O/profiles_out/
