"""Batched counterpart of the reference's tests/runtime_benchmark.py:64-157.

Times the same five phases, per batch of N envs on the GPU (HIP events around the
engine's kernels plus wall clock):
  1. full step without rendering       (runtime_benchmark.py:75-84; here: step + reset kernels)
  2. reset only                         (:90-95)
  3. physics only                       (:101-107, env.physics.step(env.state))
  4. rendering only                     (:113-130, env.observation())
  5. full step with rendering           (:136-157)

followed by the reference's six renderer settings (image size, anti_aliasing), runtime_benchmark.py:31-38
and :113-130 (`env.observation()` with a renderer of that size), with --render_sizes.

    python -m moog_demos.runtime_benchmark --config colliding_predators_32 --num_envs 4096 --render_sizes
"""
import argparse
import time

import torch

from moog import _abi, environment
from moog_demos import example_configs


def _timed(fn, reps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


def main(argv=None):
    """Runs the benchmark and prints its table; returns {'phases': {name: ms per batch}, 'render': {(size, anti_aliasing): ms per
    batch}} (what tests/test_gpu_parity.py::test_runtime_benchmark_reports_every_phase checks)."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='colliding_predators_32')
    ap.add_argument('--level', type=int, default=0)
    ap.add_argument('--num_envs', type=int, default=4096)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--render_sizes', action='store_true', help='time the six (size, anti_aliasing) renderer settings')
    ap.add_argument('--render_envs', type=int, default=256, help='batch of the renderer-size sweep')
    args = ap.parse_args(argv)
    env = environment.BatchedEnvironment(
        num_envs=args.num_envs, layer_capacity=example_configs.capacity(args.config), **example_configs.load(args.config, args.level))
    env.check_faults = False
    sync = torch.cuda.synchronize
    env.reset()
    for _ in range(5):
        env.step(env.random_action())
    n = args.num_envs
    out_img = env._out.image
    rows = []
    # 5. full step with rendering
    rows.append(('step + render', _timed(lambda: env.step(env.random_action()), args.reps, sync)))
    # 1. full step, observers disabled
    env._out.image = None
    rows.append(('step, no render', _timed(lambda: env.step(env.random_action()), args.reps, sync)))
    # 2. reset only
    rows.append(('reset only', _timed(lambda: env.reset(), max(args.reps // 5, 1), sync)))
    env._out.image = out_img
    # 3. physics only
    rows.append(('physics only', _timed(env.physics_step, args.reps, sync)))
    # 4. rendering only
    rows.append(('render only', _timed(env.observation, args.reps, sync)))
    P = env.compiled.program
    print('%s: %d envs x %d sprites, K=%d, %dx%d' % (args.config, n, P.n_slots,
                                                     P.updates_per_env_step, P.render.height,
                                                     P.render.width))
    for name, ms in rows:
        print('  %-16s %9.3f ms / batch   %12.0f env-calls/s' % (name, ms, n / ms * 1e3))
    result = {'phases': dict(rows), 'render': {}}
    if args.render_sizes:   # _IMAGE_SIZE_ANTI_ALIASING of the reference's benchmark
        from moog import observers
        env.close()
        m = args.render_envs
        for size, aa in ((64, 1), (128, 1), (256, 1), (512, 1), (512, 2), (1024, 1)):
            cfg = example_configs.load(args.config, args.level)
            old = cfg['observers']['image']
            cfg['observers'] = {'image': observers.PILRenderer(
                image_size=(size, size), anti_aliasing=aa, bg_color=old._bg_color, color_to_rgb=old.color_to_rgb,
                polygon_modifier=old.polygon_modifier)}
            e2 = environment.BatchedEnvironment(num_envs=m, layer_capacity=example_configs.capacity(args.config), **cfg)
            e2.check_faults = False
            e2.reset()
            for _ in range(3):
                e2.step(e2.random_action())
            ms = _timed(e2.observation, max(args.reps // 5, 3), sync)
            print('  render only, %4d x %-4d anti_aliasing %d: %9.3f ms / batch of %d   %10.0f frames/s' % (
                size, size, aa, ms, m, m / ms * 1e3))
            result['render'][(size, aa)] = ms
            e2.close()
    else:
        env.close()
    return result


if __name__ == '__main__':
    main()
