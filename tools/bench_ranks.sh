# Exercises bench.py's multi-rank code path on a 1-GPU box:
#  1. two ranks sharing cuda:0 over gloo (shard offsets, barrier, MAX over ranks, one JSON line)
#  2. one rank with the RCCL process group initialised (backend "nccl" init, barrier, all-reduce)
cd $GRAFT_REPO_ROOT
echo "== 2 ranks, one device, gloo"
MOOG_BENCH_ONE_DEVICE=1 MOOG_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 2>&1 | tail -3 | cut -c1-400
echo "== 2 ranks, one device, gloo, BASELINE config 5 (falling_balls_64, 8192 envs per rank)"
MOOG_BENCH_ONE_DEVICE=1 MOOG_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --workload falling_balls_64 --envs-per-gpu 8192 --steps 30 --warmup 5 \
  --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
echo "== device count as the launcher parent sees it (sysfs, no HIP)"
python -c "import bench; print(bench.visible_gpus())"
echo "== 1 rank, RCCL group"
MOOG_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 \
  python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "metric|Error|error" | cut -c1-300
