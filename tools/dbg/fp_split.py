"""first_person_predators_prey: step-kernel time with groups of rules left out (where its 1 ms per launch goes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
import torch
from moog import environment, game_rules as gr
from moog_demos import example_configs
name, n = 'first_person_predators_prey', 1024
def run(label, drop, cheap=False):
    cfg = example_configs.load(name)
    cfg['game_rules'] = tuple(r for r in cfg['game_rules'] if not drop(r))
    if cheap:   # the same rules with a one-comparison filter
        cfg['game_rules'] = tuple(gr.VanishByFilter(r._layer, lambda s: s.x < -100.) if isinstance(r, gr.VanishByFilter) else r
                                  for r in cfg['game_rules'])
    env = environment.BatchedEnvironment(num_envs=n, seed=1, layer_capacity={'prey': 32, 'predators': 96}, **cfg)
    env.check_faults = False
    env.reset()
    for _ in range(60):
        env.step(env.random_action())
    env.set_timing(True)
    for k in range(3): env.kernel_time(k)
    for _ in range(20):
        env.step(env.random_action())
    torch.cuda.synchronize()
    t = env.kernel_time(0)
    alive = int((env.state_i32[:, env.layout.o_flags:env.layout.o_flags + env.layout.S] & 1).sum().item()) / n
    print('%-34s step kernel %.0f us   live sprites / env %.1f' % (label, t[0] / max(t[1], 1) * 1e3, alive), flush=True)
    env.close()
run('all rules', lambda r: False)
run('without VanishByFilter', lambda r: isinstance(r, gr.VanishByFilter))
run('VanishByFilter, one-comparison filter', lambda r: False, cheap=True)
run('without KeepNearCenter', lambda r: isinstance(r, gr.KeepNearCenter))
run('without the ConditionalRules (CreateSprites)', lambda r: isinstance(r, gr.ConditionalRule))
run('without any rule', lambda r: True)
run('without VanishOnContact', lambda r: isinstance(r, gr.VanishOnContact))
