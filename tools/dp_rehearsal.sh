# One-GPU rehearsal of bench.py's multi-rank control flow (gloo: RCCL refuses two ranks on one device).
#   bash tools/dp_rehearsal.sh <ranks> [extra env assignments...]      e.g.  bash tools/dp_rehearsal.sh 4 GPU_MAX_HW_QUEUES=2
N=$1; shift
env "$@" VLG_BENCH_TRACE=1 VLG_BENCH_ONE_DEVICE=1 VLG_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 \
  timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29617 \
  bench.py --gpus $N --steps ${STEPS:-3} --warmup ${WARMUP:-1} --no-cpu-baseline --no-kernel-timing --no-extras
