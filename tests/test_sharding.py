"""N > 1 path on CPU: two gloo ranks, each owning a shard of the env axis, must
reproduce the single-process run bit for bit (global-index RNG keys, no data-path
collective); the timing reduction is a MAX all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from moog import sharding

N_ENVS, STEPS, NAME = 12, 6, 'functional_maze'


def test_shard_range_partitions():
    for total in (1, 7, 64, 4096, 65536):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def _run_shard(start, count, actions):
    c = helpers.compiled(NAME)
    o = helpers.OracleEnv(c, n_envs=count, seed=5, env_index0=start)
    o.reset(render=False)
    for t in range(STEPS):
        o.step(actions[t, start:start + count], render=False)
    return o.f64.copy(), o.i32.copy(), o.reward.copy()


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        actions = np.random.RandomState(0).uniform(-1, 1, size=(STEPS, N_ENVS, 2))
        start, count = sharding.shard_range(N_ENVS, rank, world)
        f, q, r = _run_shard(start, count, actions)
        # timing reduction helper: MAX over ranks
        t = sharding.max_over_ranks(1.0 + rank)
        assert t == float(world)
        # collect every shard on rank 0 (test-only gather; the hot path has no collective)
        parts = [None] * world
        dist.all_gather_object(parts, (start, f, q, r))
        if rank == 0:
            parts.sort(key=lambda p: p[0])
            ret['f'] = np.concatenate([p[1] for p in parts])
            ret['q'] = np.concatenate([p[2] for p in parts])
            ret['r'] = np.concatenate([p[3] for p in parts])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    world = 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    helpers.oracle()  # build before forking
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    actions = np.random.RandomState(0).uniform(-1, 1, size=(STEPS, N_ENVS, 2))
    f, q, r = _run_shard(0, N_ENVS, actions)
    assert np.array_equal(ret['q'], q)
    assert np.array_equal(ret['f'], f, equal_nan=True)
    assert np.array_equal(ret['r'], r, equal_nan=True)


# ---- the same with the HIP engine: two ranks (one GPU, gloo for the test's own gather) ---------------------
G_ENVS, G_STEPS, G_NAME = 96, 12, 'colliding_predators_32'


def _engine_shard(start, count, actions, device):
    import torch as th
    from moog import environment
    from moog_demos import example_configs
    env = environment.BatchedEnvironment(num_envs=count, device=device, seed=5, env_index0=start,
                                         **example_configs.load(G_NAME))
    env.reset()
    for t in range(G_STEPS):
        ts = env.step(actions[t, start:start + count])
    th.cuda.synchronize()
    out = (env.state_f64.cpu().numpy(), env.state_i32.cpu().numpy(), ts.reward.cpu().numpy(),
           ts.observation['image'].cpu().numpy())
    env.close()
    return out


def _engine_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        actions = np.random.RandomState(0).uniform(-1, 1, size=(G_STEPS, G_ENVS, 2))
        start, count = sharding.shard_range(G_ENVS, rank, world)
        f, q, r, img = _engine_shard(start, count, actions, 'cuda:0')
        assert sharding.max_over_ranks(1.0 + rank) == float(world)
        parts = [None] * world
        dist.all_gather_object(parts, (start, f, q, r, img))
        if rank == 0:
            parts.sort(key=lambda p: p[0])
            whole = _engine_shard(0, G_ENVS, actions, 'cuda:0')   # the single-engine run, same process as rank 0
            ret['same'] = all(np.array_equal(np.concatenate([p[1 + k] for p in parts]), whole[k], equal_nan=(k != 1))
                              for k in range(4))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_engine_shards_match_single_engine():
    """Two processes under torch.distributed, each driving its own engine handle over its contiguous shard
    of the env axis (env_index0 = shard start), reproduce one engine over the whole axis bit for bit:
    state records, rewards and frames.  No collective on the data path -- gloo only gathers for the check."""
    world = 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_engine_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret.get('same') is True


def test_bench_launcher_counts_devices_without_hip(tmp_path, monkeypatch):
    """`python bench.py --gpus N` (N > 1, no launcher) starts its own ranks as child processes.  The parent counts the
    GPUs from the KFD topology in sysfs (or asks a short-lived child), so it never initialises HIP itself, and it refuses
    -- exit code 2, one line on stderr -- when fewer than N are visible instead of printing a line with the wrong n_gpus."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0
    assert 'torch' not in bench.__dict__          # (the module does not import torch at load time)
    want = n + 2
    p = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', str(want), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 2, (p.returncode, p.stderr[-300:])
    assert 'only %d GPU(s) visible' % n in p.stderr
    # a visibility list narrows the count
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0')
    assert bench.visible_gpus() <= 1
