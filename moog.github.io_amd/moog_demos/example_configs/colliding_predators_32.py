"""colliding_predators scaled to 32 sprites (4 walls + 27 predators + 1 agent):
BASELINE.json configs[2], parameters from SURVEY.md 8(d) "Config 3"."""
from . import colliding_predators


def get_config(_):
    return colliding_predators.build(
        num_predators=27, predator_xy=(0.12, 0.88), predator_scale=(0.04, 0.07), agent_scale=0.05)
