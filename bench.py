"""Headline benchmark: env steps/sec of the batched MOOG step path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" = one `BatchedEnvironment.step(actions)` over the whole batch: auto-reset
kernel + step kernel (rules -> action -> K physics substeps -> task) + raster
kernel writing the uint8 frame batch, with random actions generated on device.
Workload (BASELINE.json configs[2], the one the metric is quoted on):
colliding_predators scaled to 32 sprites, 4096 envs per GPU, 64x64 observations.
For N > 1 the driver launches one rank per GPU (torch.distributed.run); envs are
independent, so ranks shard the env axis with no data-path collective (weak
scaling: 4096 envs per GPU, global env index = rank * 4096 + local).

The JSON line also carries
  roofline     achieved algorithmic GB/s of the raster kernel (HIP events around
               every launch of the timed region) against the 8 TB/s HBM3E peak
  cpu_baseline the CPU oracle (a C port of the reference algorithm) timed on this
               box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'moog.github.io_amd'))

WORKLOAD = 'colliding_predators_32'
ENVS_PER_GPU = 4096
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def raster_bytes_per_env(env):
    """Algorithmic bytes of one frame: H*W*3 written once + per live
    sprite copy nverts*2*4 B int vertices + 4 B RGBA, averaged over the batch."""
    import torch
    from moog import _abi
    P = env.compiled.program
    ncopy = 9 if P.render.polymod == _abi.MOOG_POLYMOD_TORUS else 1
    alive = env.field('alive')
    nv = env.field('nverts').to(torch.float64)
    per_env = ((nv * 8 + 4) * alive.to(torch.float64)).sum(dim=1) * ncopy
    return float(P.render.height * P.render.width * 3 + per_env.mean().item())


def raster_traffic(workload, n_envs):
    """HBM bytes per raster launch from the committed rocprofv3 PMC passes
    (profiles/raster_traffic.json; collected with tools/prof.sh), or None."""
    try:
        with open(os.path.join(REPO, 'profiles', 'raster_traffic.json')) as f:
            rec = json.load(f).get(workload)
        if rec and rec.get('n_envs') == n_envs:
            return rec['traffic_bytes']
    except (OSError, ValueError):
        pass
    return None


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU
    quota when the container has one (a 256-thread host with an 8-core quota runs 8)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:           # cgroup v2: "<quota> <period>" or "max ..."
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(seconds_target=12.0):
    """Times the CPU oracle (oracle/moog_oracle.c, one OpenMP thread per host core, envs are
    independent) on a bounded sample of the same workload: 64 envs per thread stepped until
    ~seconds_target."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import numpy as np
    import helpers
    cores = host_cores()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    n = 64 * cores
    c = helpers.compiled(WORKLOAD)
    o = helpers.OracleEnv(c, n_envs=n, seed=1)
    o.reset()
    rs = np.random.RandomState(0)
    for _ in range(2):
        o.step(rs.uniform(-1, 1, size=(n, 2)))
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds_target and steps < 400:
        o.step(rs.uniform(-1, 1, size=(n, 2)))
        steps += 1
    dt = time.perf_counter() - t0
    return {'value': n * steps / dt, 'unit': 'env steps/sec', 'cores': cores, 'kind': 'port',
            'sample': '%s, %d envs x %d steps (physics + 64x64 raster), %d OpenMP threads, %.1f s'
                      % (WORKLOAD, n, steps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--workload', default=WORKLOAD)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-schedule', action='store_true', help='disable cost-ordered launch')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from moog import _abi, environment, sharding
    from moog_demos import example_configs

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Test hooks for 1-GPU boxes (tools/bench_ranks.sh): MOOG_BENCH_ONE_DEVICE=1 puts every
    # rank on cuda:0 (with MOOG_BENCH_BACKEND=gloo, RCCL refuses two ranks on one GPU);
    # MOOG_BENCH_FORCE_DIST=1 initialises the process group even for a world of one.
    backend = os.environ.get('MOOG_BENCH_BACKEND', 'nccl')
    if os.environ.get('MOOG_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    use_dist = world > 1 or os.environ.get('MOOG_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    n = args.envs_per_gpu

    env = environment.BatchedEnvironment(
        num_envs=n, device=dev, seed=2024, env_index0=sharding.shard_range(n * world, rank, world)[0],
        **example_configs.load(args.workload))
    env.check_faults = False
    if not args.no_schedule:
        env.enable_cost_schedule()
    env.reset()
    is_grid = env._is_grid

    # random actions drawn on the device, one kernel per step, into a reused buffer
    act = torch.zeros((n,), dtype=torch.int32, device=dev) if is_grid else \
        torch.zeros((n, 2), dtype=torch.float64, device=dev)

    def one_step():
        if is_grid:
            act.random_(0, 5)
        else:
            act.uniform_(-1.0, 1.0)
        env.step(act)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_step()
    # the roofline kernel (raster) is timed live, with HIP events around every one of its
    # launches in the timed region; the other kernels' averages come from a short extra run
    # afterwards so that their event pairs do not sit in the timed stream (each pair costs
    # stream time: ~9 us around the 6 us reset kernel)
    env.set_timing(True, kernels=[_abi.MOOG_K_RASTER])
    for k in range(_abi.MOOG_K_COUNT):
        env.kernel_time(k)   # clear
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    env.set_timing(False)
    # MAX over ranks, off the timed path (RCCL needs the tensor on the GPU, gloo on the host)
    dt_max = sharding.max_over_ranks(dt, device=dev if backend == 'nccl' else None)

    k_ms = {'raster': env.kernel_time(_abi.MOOG_K_RASTER)}
    # all kernels, outside the timed region; as many steps again, because the workload is
    # periodic (episodes time out together every 200 steps: sparse right after a reset,
    # clustered contacts later), so a shorter window would not be representative
    env.set_timing(True)
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    env.set_timing(False)
    env.kernel_time(_abi.MOOG_K_RASTER)
    k_ms['step'] = env.kernel_time(_abi.MOOG_K_STEP)
    k_ms['reset'] = env.kernel_time(_abi.MOOG_K_RESET)
    faults = int((env.state_i32[:, env.layout.o_fault] != 0).sum().item())
    if rank == 0:
        total_steps = n * world * args.steps
        rb = raster_bytes_per_env(env)
        r_ms, r_n = k_ms['raster']
        r_avg_s = (r_ms / max(r_n, 1)) * 1e-3
        achieved = (n * rb / r_avg_s) / 1e9 if r_avg_s > 0 else 0.0
        P = env.compiled.program
        line = {
            'metric': 'env steps/sec (whole node), 4096 envs x 32 sprites, 64x64 obs',
            'value': total_steps / dt_max,
            'unit': 'env steps/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt_max / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '%s: %d envs/GPU x %d sprites, K=%d substeps, %dx%d raster, '
                                   'random joystick actions, auto-reset on' % (
                                       args.workload, n, P.n_slots, P.updates_per_env_step,
                                       P.render.height, P.render.width),
                       'envs_per_gpu': n, 'sprites': P.n_slots, 'obs': [P.render.height, P.render.width],
                       'parallelism': 'env-sharded x%d, no collective' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': raster_traffic(args.workload, n),
                         'traffic_unit': 'bytes per launch (rocprofv3 PMC, profiles/r01_current.txt)',
                         'algorithmic_bytes_per_launch': n * rb,
                         'kernel': 'moog_raster_kernel', 'avg_kernel_us': r_avg_s * 1e6,
                         'algorithmic_bytes_per_env': rb},
            'kernels_avg_us': {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in k_ms.items()},
            'faulted_envs': faults,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
