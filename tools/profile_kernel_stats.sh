# rocprofv3 kernel-trace summaries of bench.py (f32, bf16) and of the reference step -> gpurun_out/prof_*; run on the GPU box:
#   gpurun -- bash tools/profile_kernel_stats.sh r02_a     (copies <tag>_*_kernel_stats.csv + bench lines to gpurun_out/profiles_out/)
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_f32 $O/prof_bf16 $O/prof_ref $O/prof_shard
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_f32.json 2> $O/prof_f32.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > $O/prof_bf16.json 2> $O/prof_bf16.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref -- python $R/tools/reference_step_bench.py 4 1 > $O/prof_ref.log 2> $O/prof_ref.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -- python $R/tools/shard_bench.py 4 50 > $O/prof_shard.log 2> $O/prof_shard.err
mkdir -p $O/profiles_out
cp $(find $O/prof_shard -name "*kernel_stats.csv" | head -1) $O/profiles_out/${TAG}_shard4_kernel_stats.csv; cp $O/prof_shard.log $O/profiles_out/${TAG}_shard4.log
for m in f32 bf16; do cp $(find $O/prof_$m -name "*kernel_stats.csv" | head -1) $O/profiles_out/${TAG}_${m}_kernel_stats.csv; cp $O/prof_$m.json $O/profiles_out/${TAG}_${m}_bench.json; done
cp $(find $O/prof_ref -name "*kernel_stats.csv" | head -1) $O/profiles_out/${TAG}_reference_step_kernel_stats.csv
