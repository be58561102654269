# rocprofv3 kernel-trace summaries of bench.py (f32, bf16) and of the reference step -> gpurun_out/prof_*; run on the GPU box:
#   gpurun -- bash tools/profile_kernel_stats.sh     (then copy the *_kernel_stats.csv files into profiles/)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_f32.json 2> $O/prof_f32.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > $O/prof_bf16.json 2> $O/prof_bf16.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref -- python /root/repo/tools/reference_step_bench.py 4 1 > $O/prof_ref.log 2> $O/prof_ref.err
