cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_bf16 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > /root/repo/gpurun_out/prof_bf16.json 2> /root/repo/gpurun_out/prof_bf16.err
