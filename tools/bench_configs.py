"""Per-config throughput of the engine (BASELINE.json configs 2-5 and the other lowered configs), kernel times:
reset, 5 warm-up calls, then `steps` timed calls with random actions (HIP-event kernel times of those calls)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moog.github.io_amd'))
import torch
from moog import environment, _abi
from moog_demos import example_configs

# capacities of the layers rules append to, sized from env.layer_usage() (the recipes' own LAYER_CAPACITY values are the
# ones the reference fixtures were recorded with and are too small for a 4096-env batch's tail)
BENCH_CAPACITY = {'first_person_predators_prey': {'prey': 32, 'predators': 96}, 'rules_zoo_l1': {'prey': 24, 'predators': 24}}

def run(name, n, steps=30, observers=True, **kw):
    cfg = example_configs.load(name) if not kw else __import__('moog_demos.example_configs.' + name, fromlist=['x']).get_config(0, **kw)
    env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity=BENCH_CAPACITY.get(name, example_configs.capacity(name)), **cfg)
    env.check_faults = False
    env.reset()
    for _ in range(5):
        env.step(env.random_action())
    env.set_timing(True)
    for k in range(3): env.kernel_time(k)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        env.step(env.random_action())
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    ks = {nm: env.kernel_time(k) for nm, k in (('step', 0), ('raster', 1), ('reset', 2))}
    faults = int((env.state_i32[:, env.layout.o_fault] != 0).sum().item())
    print('%-24s N=%6d  %10.0f env-steps/s  step %.0f us  raster %.0f us  reset %.0f us  faults %d' % (
        name, n, n * steps / dt, *(ks[k][0] / max(ks[k][1], 1) * 1e3 for k in ('step', 'raster', 'reset')), faults), flush=True)
    use = env.layer_usage()
    if use:
        print('    dynamic layers:', use, flush=True)

run('chase_avoid_torus', 4096)
run('colliding_predators_32', 4096)
run('functional_maze', 8192, image_size=(128, 128))
run('falling_balls_64', 8192, steps=10)
run('pong', 4096)
run('colliding_predators', 4096)
run('falling_balls', 4096)
run('first_person_predators_prey', 4096, steps=60)
run('lambda_zoo', 4096)
run('rules_zoo_l1', 4096)
run('tether_zoo_l0', 4096)
run('distrib_zoo', 4096)
run('cleanup', 4096, steps=60)
run('maze_zoo', 4096, steps=60)
run('pacman', 1024, steps=60)
run('pacman', 4096, steps=60)
# the reference configs unlocked in rounds 2 and 3 (their own files load unchanged)
run('parallelogram_catch', 4096, steps=60)
run('multi_tracking_with_feature_l3', 4096, steps=60)
run('match_to_sample_l3', 4096, steps=60)
run('predators_arena_l2', 4096, steps=60)
run('bounce_box_contact_prediction', 1024, steps=60)   # (a reset plays the episode forward: ~200 physics steps inside it)
run('red_green_l1', 1024, steps=60)                    # (likewise, and rejects unusable trials)
