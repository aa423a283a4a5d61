"""falling_balls scaled to 64 sprites (4 walls + 60 balls): BASELINE.json
configs[4], parameters from SURVEY.md 8(d) "Config 5"."""
from . import falling_balls


def get_config(_):
    return falling_balls.build(num_balls=60, x_range=(0.1, 0.9), y_range=(0.35, 1.05), scale=0.04)
