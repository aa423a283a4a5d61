"""Per-config throughput of the engine (BASELINE.json configs 2-5 and the other lowered configs), kernel times:
reset, 5 warm-up calls, then `steps` timed calls with random actions (HIP-event kernel times of those calls)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')   # (the reset pool's fills and nothing else here want more than the runtime's four; set before HIP initialises)
from moog_demos import example_configs

# capacities of the layers rules append to, sized from env.layer_usage() (the recipes' own LAYER_CAPACITY values are the
# ones the reference fixtures were recorded with and are too small for a 4096-env batch's tail)
BENCH_CAPACITY = {'first_person_predators_prey': {'prey': 32, 'predators': 96}, 'rules_zoo_l1': {'prey': 24, 'predators': 24}}

def config_of(name, kw):
    kw = {k: v for k, v in kw.items() if k not in ('steps', 'capacity', 'fit', 'warm')}
    return example_configs.load(name) if not kw else __import__('moog_demos.example_configs.' + name, fromlist=['x']).get_config(0, **kw)


def capacity_of(name, kw):
    return kw.get('capacity') or BENCH_CAPACITY.get(name, example_configs.capacity(name))


def run(name, n, steps=30, observers=True, **kw):
    import torch
    from moog import environment
    cfg = config_of(name, kw)
    env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=capacity_of(name, kw), **cfg)
    env.check_faults = False
    env.reset()
    if kw.get('fit'):   # size the appendable layers by what the batch needs (BatchedEnvironment.fit_layer_capacity) after a warm-up
        for _ in range(int(kw['fit'])):
            env.step(env.random_action())
        print('    fit_layer_capacity:', env.fit_layer_capacity(), flush=True)
        env.check_faults = False
    for _ in range(int(kw.get('warm', 5))):
        env.step(env.random_action())
    env.check_faults = False
    env.set_timing(True)
    for k in range(3): env.kernel_time(k)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        env.step(env.random_action())
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    ks = {nm: env.kernel_time(k) for nm, k in (('step', 0), ('raster', 1), ('reset', 2))}
    faults = int((env.state_i32[:, env.layout.o_fault] != 0).sum().item())
    print('%-24s N=%6d  %10.0f env-steps/s  step %.0f us  raster %.0f us  reset %.0f us  faults %d  [%s]' % (
        name, n, n * steps / dt, *(ks[k][0] / max(ks[k][1], 1) * 1e3 for k in ('step', 'raster', 'reset')), faults, env.step_kernel()), flush=True)
    use = env.layer_usage()
    if use:
        print('    dynamic layers:', use, flush=True)

CASES = [
    ('chase_avoid_torus', 4096, dict()),
    ('colliding_predators_32', 4096, dict()),
    ('functional_maze', 8192, dict(image_size=(128, 128))),
    ('falling_balls_64', 8192, dict(steps=10)),
    ('pong', 4096, dict()),
    ('colliding_predators', 4096, dict()),
    ('falling_balls', 4096, dict()),
    ('first_person_predators_prey', 4096, dict(steps=60)),
    # (the same with the layers sized to their high-water marks -- 26 prey, 41 predators at 4096 envs: the record shrinks from 85 to
    #  55 KB and two envs share a CU's LDS instead of one)
    ('first_person_predators_prey', 4096, dict(steps=60, capacity={'prey': 32, 'predators': 48})),
    # (and as a user gets it by default: layer_capacity 'auto' fits the layers by itself after 128 calls and grows them on demand)
    ('first_person_predators_prey', 4096, dict(steps=60, warm=140, capacity={'auto': True, 'prey': 32, 'predators': 96})),
    # (fit=<calls>: the same sized by the engine itself after a warm-up, BatchedEnvironment.fit_layer_capacity(); not in the table: with a
    #  random policy this config's prey pile up, the layers keep growing and a 60-call window mostly times engine re-creations)
    ('lambda_zoo', 4096, dict()),
    ('rules_zoo_l1', 4096, dict()),
    ('tether_zoo_l0', 4096, dict()),
    ('distrib_zoo', 4096, dict()),
    ('cleanup', 4096, dict(steps=60)),
    ('maze_zoo', 4096, dict(steps=60)),
    ('pacman', 1024, dict(steps=60)),
    ('pacman', 4096, dict(steps=60)),
    # the reference configs unlocked in rounds 2 and 3 (their own files load unchanged)
    ('parallelogram_catch', 4096, dict(steps=60)),
    ('multi_tracking_with_feature_l3', 4096, dict(steps=60)),
    ('match_to_sample_l3', 4096, dict(steps=60)),
    ('predators_arena_l2', 4096, dict(steps=60)),
    ('bounce_box_contact_prediction', 1024, dict(steps=60)),   # (a reset plays the episode forward: ~200 physics steps inside it)
    ('red_green_l1', 1024, dict(steps=60)),                    # (likewise, and rejects unusable trials)
]


if __name__ == '__main__':
    if '--build-spec' in sys.argv:   # no GPU: the specialised step kernel of every case's program (moog/_spec.py), in parallel
        import concurrent.futures
        from moog import _compiler, _spec
        def plain(cap):   # ('auto' / 'fit_after' are BatchedEnvironment's: the program is the one of the initial capacities)
            return {k: v for k, v in cap.items() if k not in ('auto', 'fit_after')} if isinstance(cap, dict) else cap
        progs = [(name, _compiler.compile_config(layer_capacity=plain(capacity_of(name, kw)),
                                                 **config_of(name, kw)).program) for name, _, kw in CASES]
        with concurrent.futures.ThreadPoolExecutor(max_workers=7) as ex:
            for (name, _), path in zip(progs, ex.map(lambda p: _spec.build(p[1]), progs)):
                print('%-32s %s' % (name, os.path.basename(path)), flush=True)
        sys.exit(0)
    only = [a for a in sys.argv[1:] if not a.startswith('-')]
    for name, n, kw in CASES:
        if not only or name in only:
            run(name, n, **kw)
