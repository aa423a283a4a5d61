"""Headline benchmark: env steps/sec of the batched MOOG step path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" = one `BatchedEnvironment.step(actions)` over the whole batch: step
kernel (auto-reset of the envs whose episode ended, else rules -> action -> K
physics substeps -> task) + raster kernel writing the uint8 frame batch, with
random actions generated on the device before the timed region.
Workload (BASELINE.json configs[2], the one the metric is quoted on):
colliding_predators scaled to 32 sprites, 4096 envs per GPU, 64x64 observations.
`--workload falling_balls_64 --envs-per-gpu 8192` is BASELINE.json configs[4]
(the 8-GPU weak-scaling stress).

N > 1: one process per GPU.  Launched under torch.distributed.run (the driver's
form, WORLD_SIZE set) the script is one rank; launched plainly with `--gpus N`
it starts the N ranks itself (before anything touches a GPU) and relays rank
0's JSON line.  Envs are independent, so ranks shard the env axis with no
data-path collective (weak scaling: global env index = rank * envs_per_gpu +
local); one MAX all-reduce of the wall time is the only communication.

Stationary load: every episode of the headline workload lasts exactly
`timeout_steps` calls, so a batch that was reset together would time out
together and the cost of a step would depend on where in the episode the timed
window falls.  The benchmark therefore de-synchronises the episodes before
timing (each env starts at a different `step_count`) and burns in one episode
length; after that the same fraction of the batch is resetting in every call,
as in any long-running job.  `--lockstep` keeps the synchronous episodes.

The JSON line also carries
  roofline     achieved algorithmic GB/s of the raster kernel (HIP events around
               every launch of the timed region) against the 8 TB/s HBM3E peak
  cpu_baseline the CPU oracle (a C port of the reference algorithm) timed on this
               box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import socket
import subprocess
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


def raster_kernel_name(env):
    """Which rasteriser draws the frames (moog_engine_raster_path): the name a rocprofv3 trace of this run shows."""
    part = env.parts[0] if hasattr(env, 'parts') else env
    words = 2 if part.compiled.program.render.width > 64 else 1
    if part.raster_path() == 'mask':
        P = part.compiled.program
        big = any(P.slot_vcap[s] > 32 for s in range(P.n_slots))   # (the instantiation with the long-polygon row routine)
        compact = part.raster_compact_edges()                      # (4-byte edge records: programs whose 16-byte ones keep frames off a CU)
        return 'moog_raster_mask_kernel<%d, %s, %s> (csrc/moog_raster_mask_core.h)' % (words, 'true' if big else 'false', 'true' if compact else 'false')
    return 'moog_raster_kernel<%d> (csrc/moog_raster_kernel.h)' % words


def raster_traffic(workload, n_envs):
    """HBM bytes per raster launch as profiled offline (rocprofv3 PMC passes of this
    workload, tools/prof.sh -> profiles/raster_traffic.json; the record names the
    profile file and commit it came from), or None.  Not a measurement of this run."""
    try:
        with open(os.path.join(REPO, 'profiles', 'raster_traffic.json')) as f:
            rec = json.load(f).get(workload)
        if rec and rec.get('n_envs') == n_envs:
            return rec
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


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _time_oracle(workload, n, threads, seconds_target, max_steps, phase='full'):
    import ctypes
    import numpy as np
    import helpers
    try:   # the oracle is OpenMP over envs: pin the team size for this sample
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(int(threads))
    except OSError:
        pass
    c = helpers.compiled(workload)
    o = helpers.OracleEnv(c, n_envs=n, seed=1)
    o.reset()
    rs = np.random.RandomState(0)
    grid = bool(c.program.action.kind == 2 and c.program.n_actions <= 1)
    act = (lambda: rs.randint(0, 5, size=n)) if grid else (lambda: rs.uniform(-1, 1, size=(n, 2)))
    for _ in range(2):
        o.step(act())
    # (the phases without frames: the oracle's step with rendering off; its physics-only entry point is not timed apart)
    one = {'full': lambda: o.step(act()), 'render': o.render}.get(phase, lambda: o.step(act(), render=False))
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds_target and steps < max_steps:
        one()
        steps += 1
    dt = time.perf_counter() - t0
    return n * steps / dt, steps, dt


def cpu_baseline(workload, seconds_target=10.0, phase='full'):
    """Times the CPU oracle (oracle/moog_oracle.c, a C restatement of the reference
    algorithm) on a bounded sample of the same workload: first one thread (the
    reference's own single-threaded design), then one OpenMP thread per host core over
    disjoint env shards (envs are independent)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    cores = host_cores()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    v1, s1, d1 = _time_oracle(workload, 32, 1, seconds_target * 0.5, 200, phase)
    vn, sn, dn = _time_oracle(workload, 64 * cores, cores, seconds_target, 400, phase)
    return {'value': vn, 'unit': 'env steps/sec', 'cores': cores, 'kind': 'port',
            'cpu_model': cpu_model(),
            'single_thread': {'value': v1, 'unit': 'env steps/sec', 'cores': 1,
                              'sample': '%s, 32 envs x %d steps, 1 thread, %.1f s' % (workload, s1, d1)},
            'sample': '%s, %d envs x %d steps (%s), %d OpenMP threads, %.1f s'
                      % (workload, 64 * cores, sn, {'full': 'physics + raster', 'render': 'raster only'}.get(phase, 'step without raster'),
                         cores, dn)}


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs this process tree may use, counted WITHOUT initialising HIP in this process (the launcher parent must stay
    clean: its children are the ones that open the devices).  First the KFD topology in sysfs -- a node with SIMDs is a
    GPU -- filtered by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are plain index lists; if sysfs is not
    readable, a short-lived child process asks the runtime (torch.cuda.device_count()) and prints the answer."""
    n = None
    try:
        root = '/sys/class/kfd/kfd/topology/nodes'
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    if n is not None and n > 0:
        opaque = False   # a *_VISIBLE_DEVICES list of UUIDs (GPU-xxxx): sysfs cannot tell which nodes it names
        for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
            v = os.environ.get(var)
            if v is None:
                continue
            toks = [t.strip() for t in v.split(',') if t.strip()]
            if all(t.isdigit() for t in toks):
                n = min(n, len(toks))
            else:
                n = min(n, len(toks))
                opaque = True
        if not opaque:
            return n
        # (sysfs also lists GPUs this container may not open: with an opaque filter ask a child what the runtime really sees)
    try:   # (a child: whatever it initialises dies with it)
        out = subprocess.check_output([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                                      stderr=subprocess.DEVNULL, timeout=300)
        seen = int(out.decode().strip().splitlines()[-1])
        return seen if not n else min(n, seen)
    except (subprocess.SubprocessError, ValueError, IndexError):
        return 0


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run as child
    processes.  This process never touches the HIP runtime (visible_gpus reads sysfs or asks a child), and nothing is
    re-executed: the ranks are children, this process relays their exit code."""
    have = visible_gpus()
    if have < args.gpus:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) visible\n' % (args.gpus, have))
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def stagger_episodes(env, timeout, index0):
    """De-synchronises the episodes: env i starts its first episode at step_count
    (i * 7919) mod timeout (7919 is prime: a contiguous shard of the env axis covers all
    episode phases evenly)."""
    import torch
    n = env.num_envs
    idx = torch.arange(n, dtype=torch.int64, device=env.device) + int(index0)
    env.state_i32[:, env.layout.o_step_count] = ((idx * 7919) % int(timeout)).to(torch.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--total-envs', type=int, default=0,
                    help='strong scaling: this many envs in all, split over the ranks (shard_range); overrides --envs-per-gpu')
    ap.add_argument('--workload', default=WORKLOAD,
                    help='a recipe of moog_demos.example_configs; name@size = the recipe with a size x size renderer '
                         '(functional_maze@128 --envs-per-gpu 8192 is BASELINE.json configs[3])')
    ap.add_argument('--phase', default='full', choices=('full', 'step', 'physics', 'render'),
                    help="what a timed step is (the phases of the reference's tests/runtime_benchmark.py:64-157): full = step + frames "
                         '(the headline); step = the full step with the observers disabled (:75-84; BASELINE.json configs[1], '
                         '"physics-only (no observer)", is --workload chase_avoid_torus --phase step); physics = '
                         'env.physics.step alone (:101-107, moog_engine_physics_only); render = env.observation() alone (:113-130, '
                         'moog_engine_render)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-schedule', action='store_true', help='disable cost-ordered launch')
    ap.add_argument('--no-fused', action='store_true', help='(accepted and ignored: the launch structure it switched off was retired with ABI 30)')
    ap.add_argument('--lockstep', action='store_true', help='keep the episodes of the batch synchronous')
    ap.add_argument('--no-extras', action='store_true', help='skip the strict-fault-check comparison window')
    ap.add_argument('--no-spec', action='store_true',
                    help='step with the generic kernels even if one specialised for the program exists (moog/_spec.py)')
    ap.add_argument('--sub-batches', type=int, default=1,
                    help='G > 1: the batch is stepped as G asynchronous sub-batches, one HIP stream each '
                         '(moog.environment.SubBatchedEnvironment): an ADDITIONAL line, the synchronous whole-batch '
                         'step is the headline')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)

    # Hardware queues of this process (read by the HIP runtime when it initialises; its default is 4): the asynchronous
    # sub-batches run beside each other only on queues of their own, two per sub-batch.  Set here, by the benchmark, and
    # reported in the JSON line -- the engine library leaves the variable alone.
    if args.sub_batches > 1:
        os.environ.setdefault('GPU_MAX_HW_QUEUES', str(max(4, 2 * args.sub_batches)))
    if args.no_spec:
        os.environ['MOOG_STEP_SPEC'] = '0'
    import torch
    import torch.distributed as dist
    from moog import _abi, environment, sharding
    from moog_demos import example_configs

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Test hooks for 1-GPU boxes (tools/bench_ranks.sh): MOOG_BENCH_ONE_DEVICE=1 puts every
    # rank on cuda:0 (with MOOG_BENCH_BACKEND=gloo, RCCL refuses two ranks on one GPU);
    # MOOG_BENCH_FORCE_DIST=1 initialises the process group even for a world of one.
    backend = os.environ.get('MOOG_BENCH_BACKEND', 'nccl')
    if os.environ.get('MOOG_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write('bench.py: rank %d has no GPU (%d visible)\n' % (rank, torch.cuda.device_count()))
        sys.exit(2)
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
    if args.total_envs > 0:   # strong scaling: a fixed batch, split as evenly as the env axis allows
        index0, n = sharding.shard_range(args.total_envs, rank, world)
    else:
        n = args.envs_per_gpu
        index0 = sharding.shard_range(n * world, rank, world)[0]

    G = max(1, args.sub_batches)
    if G > 1:
        env = environment.SubBatchedEnvironment(
            num_envs=n, sub_batches=G, device=dev, seed=2024, env_index0=index0,
            layer_capacity=example_configs.capacity(args.workload),
            **example_configs.load(args.workload))
        args.no_extras = True
    else:
        # (the step kernel specialised for this program is built by __graft_entry__.build(); if it is not there -- another
        #  workload, a fresh checkout -- it is compiled here, before anything is timed: ~20 s of hipcc, no GPU work)
        spec = False
        if os.environ.get('MOOG_SPEC_PREBUILT'):   # (A/B runs: kernels built by hand in MOOG_SPEC_DIR are used as they are)
            spec = True
        elif not args.no_spec:
            # (rank 0 builds, the others wait for it: eight hipcc runs at once would only slow each other down)
            if rank == 0:
                try:
                    from moog import _compiler, _spec
                    _spec.build(_compiler.compile_config(layer_capacity=example_configs.capacity(args.workload),
                                                         **example_configs.load(args.workload)).program)
                    spec = True
                except Exception as exc:   # pylint: disable=broad-except  (no hipcc on the box: the generic kernels do)
                    sys.stderr.write('bench.py: no specialised step kernel (%s)\n' % (exc,))
            if use_dist:
                dist.barrier()
        env = environment.BatchedEnvironment(
            num_envs=n, device=dev, seed=2024, env_index0=index0,
            layer_capacity=example_configs.capacity(args.workload),
            **example_configs.load(args.workload))
    if not args.no_schedule:
        env.enable_cost_schedule()
    env.reset()
    is_grid = env._is_grid
    P = env.compiled.program
    timeout = P.timeout_steps
    staggered = (not args.lockstep) and timeout == timeout and 1 < timeout < 1e6

    # Synthetic input: random actions drawn on the device BEFORE the timed region (the inputs are resident in HBM
    # when it starts): a ring of action batches, one per step of the window (at most 1024, then it wraps).
    ring = max(1, min(1024, args.steps))
    if is_grid:
        acts = torch.randint(0, 5, (ring, n), dtype=torch.int32, device=dev)
    else:
        acts = torch.empty((ring, n, 2), dtype=torch.float64, device=dev).uniform_(-1.0, 1.0)
    step_no = [0]

    m = n // G

    phase = ['full']   # (burn-in and warm-up of the physics / render phases are full steps: the timed phase starts from the stationary mix)

    def one_step():
        a = acts[step_no[0] % ring]
        if G > 1:   # every sub-batch queues its step behind its own previous call only: no whole-batch barrier
            for g in range(G):
                env.step_async(g, a[g * m:(g + 1) * m])
        elif phase[0] == 'physics':
            env.physics_step()
        elif phase[0] == 'render':
            env.observation()
        else:
            env.step(a)
        step_no[0] += 1

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    burn_in = 0
    if staggered:   # setup, not part of the W warm-up steps: reach the stationary mix of episode phases
        stagger_episodes(env, timeout, index0)
        burn_in = int(timeout) + 1
        for _ in range(burn_in):
            one_step()
    if args.phase != 'full' and G > 1:
        sys.stderr.write('bench.py: --phase %s needs --sub-batches 1\n' % args.phase)
        sys.exit(2)
    if args.phase == 'step':   # observers disabled: the engine is handed no image pointer (moog_demos/runtime_benchmark.py phase 1)
        env._out.image = None
        args.no_extras = True
    phase[0] = args.phase
    if args.phase in ('physics', 'render'):
        args.no_extras = True
    for _ in range(args.warmup):
        one_step()
    # Kernels of the timed region are bracketed by HIP events on the launch stream, so the per-kernel
    # averages (and the roofline figure of the raster kernel) are measurements of the timed steps
    # themselves.  Every 8th launch of each kernel is bracketed (25 samples per kernel in the default window,
    # never fewer than 10): bracketing every launch costs the timed region 20 us per step (2 %), measured.
    every = max(1, min(8, args.steps // 10))
    env.set_timing(True, every=every)
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
    rdev = dev if backend == 'nccl' else None
    dt_max = sharding.max_over_ranks(dt, device=rdev)
    # what the collective itself saw: ranks that took part, envs they stepped, the slowest and the fastest rank's own rate
    ranks_seen, envs_seen = sharding.reduce_over_ranks([1.0, float(n)], 'sum', device=rdev)
    rate_min = sharding.reduce_over_ranks([n * args.steps / dt], 'min', device=rdev)[0]
    rate_max = sharding.reduce_over_ranks([n * args.steps / dt], 'max', device=rdev)[0]
    env.raise_faults()   # faults of the timed steps (deferred surfacing: none were skipped silently)

    k_ms = {'raster': env.kernel_time(_abi.MOOG_K_RASTER), 'step': env.kernel_time(_abi.MOOG_K_STEP),
            'reset': env.kernel_time(_abi.MOOG_K_RESET)}
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        # the same steps with a host synchronisation + fault check after every call
        # (check_faults = 'sync': exceptions surface in the call that caused them)
        env.check_faults = 'sync'
        m = max(20, args.steps // 4)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(m):
            one_step()
        torch.cuda.synchronize(dev)
        extras['value_sync_fault_check'] = n * m / (time.perf_counter() - t1)
        env.check_faults = True
    faults = int((env.state_i32[:, env.layout.o_fault] != 0).sum().item())
    if rank == 0:
        total_steps = int(envs_seen) * args.steps
        frames = args.phase in ('full', 'render')
        if frames:
            rb = raster_bytes_per_env(env)
            r_ms, r_n = k_ms['raster']
            roof_kernel = raster_kernel_name(env)
            tr = raster_traffic(args.workload, n)
        else:
            # No frames in this phase: the launch that dominates is the step kernel.  Its algorithmic bytes (SURVEY 8d B_phys: the
            # mutable sprite state read and written once, the per-sprite parameters read, the action; sub-steps stay on chip):
            # S x (6 x 8 + 6 x 8 + 1) + S x 56 + 16 bytes per env-step.  It is a latency-bound O(S^2) geometry kernel: the HBM
            # fraction is reported because the contract asks for one, not because HBM bounds it.
            rb = float(P.n_slots * 153 + 16)
            r_ms, r_n = k_ms['step']
            roof_kernel = 'moog_step_kernel (csrc/moog_kernels.h; not HBM-bound: one wavefront per env, dependent fp64 chain)'
            tr = None
        r_avg_s = (r_ms / max(r_n, 1)) * 1e-3
        achieved = (n * rb / r_avg_s) / 1e9 if r_avg_s > 0 else 0.0
        line = {
            'metric': 'env steps/sec (whole node), %d envs x %d sprites, %dx%d obs' % (
                n, P.n_slots, P.render.height, P.render.width) + ('' if args.phase == 'full' else ' [phase: %s]' % args.phase),
            'value': total_steps / dt_max,
            'unit': 'env steps/sec',
            'n_gpus': world,
            'ranks_seen': int(ranks_seen),
            'per_rank_value': {'min': rate_min, 'max': rate_max},
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt_max / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'strong' if args.total_envs > 0 else 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '%s: %d envs/GPU x %d sprites, K=%d substeps, %dx%d raster, '
                                   'random joystick actions, auto-reset on' % (
                                       args.workload, n, P.n_slots, P.updates_per_env_step,
                                       P.render.height, P.render.width),
                       'phase': {'full': 'step + frames', 'step': 'full step, observers disabled (no frames)',
                                 'physics': 'physics.step only (moog_engine_physics_only)',
                                 'render': 'env.observation() only (moog_engine_render)'}[args.phase],
                       'envs_per_gpu': n, 'total_envs': int(envs_seen), 'sprites': P.n_slots, 'obs': [P.render.height, P.render.width],
                       'parallelism': 'env-sharded x%d, no collective' % world,
                       'launch': (('%d asynchronous sub-batches of %d envs, one HIP stream each: step -> frames -> next step '
                                   'chained per sub-batch, no whole-batch barrier between calls (SubBatchedEnvironment.step_async)'
                                   % (G, m)) if G > 1 else 'step launch, then raster launch, on one stream'),
                       'sub_batches': G,
                       'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'runtime default (4)'),
                       'step_kernel': ((env.parts[0] if hasattr(env, 'parts') else env).step_kernel() +
                                       ' (moog_step_kernel compiled with the program as a constant, moog/_spec.py)'
                                       if (env.parts[0] if hasattr(env, 'parts') else env).step_kernel() == 'specialised'
                                       else 'generic (moog_step_kernel reading the program at run time)'),
                       'episodes': ('staggered (step_count offsets + %d burn-in steps before the warm-up)' % burn_in)
                                   if staggered else 'lockstep'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS,
                         'traffic': tr['traffic_bytes'] if tr else None,
                         'traffic_source': (tr.get('source', 'profiles/raster_traffic.json') + ' (offline rocprofv3 PMC passes, '
                                            'not this run)') if tr else None,
                         'algorithmic_bytes_per_launch': n * rb,
                         'kernel': roof_kernel, 'avg_kernel_us': r_avg_s * 1e6, 'kernel_samples': int(r_n),
                         'algorithmic_bytes_per_env': rb},
            'kernels_avg_us': {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in k_ms.items() if v[1] > 0},
            'kernels_avg_us_note': 'HIP-event brackets around every %dth launch inside the timed region' % every,
            'faulted_envs': faults,
        }
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.workload, phase=args.phase)
        elif world > 1:
            line['cpu_baseline'] = None
            line['cpu_baseline_note'] = 'timed by the --gpus 1 run only (rank 0, N = 1)'
        print(json.dumps(line))
        sys.stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
