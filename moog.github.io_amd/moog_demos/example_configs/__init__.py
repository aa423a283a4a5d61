"""Task recipes for the batched engine.

Each module exposes `get_config(level) -> dict` whose keys are the kwargs of
`moog.environment.Environment` -- the same "plugin API" as the reference's
`moog_demos/example_configs/*.py`.  The five recipes named in BASELINE.json are
re-stated here (parameters cited per file) so that tests and the benchmark can
run on machines where the reference checkout is absent; the reference's own
config files also load unchanged (tests/test_configs.py).  `*_32` / `*_64` are
the scaled variants of SURVEY.md 8(d).
"""
import importlib
import re

NAMES = ('pong', 'chase_avoid_torus', 'colliding_predators', 'functional_maze', 'falling_balls',
         'first_person_predators_prey', 'cleanup', 'pacman', 'parallelogram_catch', 'multi_tracking_with_feature_l3', 'match_to_sample_l3', 'predators_arena_l2', 'bounce_box_contact_prediction', 'red_green_l1',
         'colliding_predators_32', 'falling_balls_64', 'forces_zoo', 'tether_zoo', 'distrib_zoo', 'rules_zoo', 'lambda_zoo', 'cond_zoo', 'phase_zoo', 'actions_zoo', 'aa_zoo', 'maze_zoo', 'sampler_zoo', 'dependent_zoo', 'lookahead_zoo', 'tracing_zoo', 'combo_zoo')


def capacity(name):
    """layer_capacity for BatchedEnvironment / compile_config: run-time sprite capacity of the
    layers a recipe's rules append to (module attribute LAYER_CAPACITY), or None."""
    name = name.partition('@')[0]
    m = re.match(r'(.*)_l(\d+)$', name)
    if m:
        name = m.group(1)
    return getattr(importlib.import_module(__name__ + '.' + name), 'LAYER_CAPACITY', None)


def load(name, level=0):
    """`name`: a recipe module; `<name>_l<k>`: its level k; `<name>@<size>`: the recipe with a size x size renderer
    (functional_maze@128 = BASELINE.json configs[3]; the recipe's get_config must take `image_size`)."""
    name, _, size = name.partition('@')
    m = re.match(r'(.*)_l(\d+)$', name)   # e.g. chase_avoid_torus_l1 = level 1 of chase_avoid_torus
    if m:
        name, level = m.group(1), int(m.group(2))
    get_config = importlib.import_module(__name__ + '.' + name).get_config
    if size:
        return get_config(level, image_size=(int(size), int(size)))
    return get_config(level)
