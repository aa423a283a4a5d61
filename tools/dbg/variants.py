"""Which step kernel each lowered config runs on, and whether its episodes are opened by the late reset: python tools/dbg/variants.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment
from moog_demos import example_configs
for name in ('pong', 'chase_avoid_torus', 'colliding_predators', 'colliding_predators_32', 'functional_maze', 'falling_balls', 'falling_balls_64',
             'first_person_predators_prey', 'cleanup', 'pacman', 'parallelogram_catch', 'multi_tracking_with_feature_l3', 'match_to_sample_l3',
             'predators_arena_l2', 'bounce_box_contact_prediction', 'red_green_l1', 'maze_zoo', 'lambda_zoo', 'rules_zoo_l1', 'callables_zoo',
             'lookahead_zoo', 'tracing_zoo', 'combo_zoo', 'dependent_zoo', 'distrib_zoo'):
    try:
        env = environment.BatchedEnvironment(num_envs=8, seed=1, layer_capacity=example_configs.capacity(name), **example_configs.load(name))
        print('%-34s variant %d  late reset %s  pool %s' % ((name,) + env.kernel_variant + (env.reset_pool['on'],)))
        env.close()
    except Exception as ex:   # pylint: disable=broad-except
        print('%-34s %r' % (name, ex))
