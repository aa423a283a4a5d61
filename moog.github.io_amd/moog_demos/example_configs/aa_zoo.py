"""Coverage recipe (not a reference task): the renderer beyond the BASELINE settings -- anti-aliased canvases
(`anti_aliasing` > 1: the frame is drawn `anti_aliasing` times larger and Image.resize(LANCZOS)-ed down,
reference moog/observers/pil_renderer.py:64-66,111-112), non-square observations, translucent sprites on a
coloured background -- pinned by golden vectors captured from the reference (tests/golden/aa_zoo_*.npz).
level 0: 64 x 48 observation, anti_aliasing 3;  level 1: 32 x 32, anti_aliasing 2, TorusGeometry;
level 2: 128 x 128, anti_aliasing 2 (a 256 x 256 canvas: several tiles of the rasteriser);
levels 3-5: sizes that are no multiples of 16 or 4 (pil_renderer.py:64-66 takes any size): 50 x 37, anti_aliasing 1;
30 x 22, anti_aliasing 3 (a 90 x 66 canvas); 150 x 41, anti_aliasing 1, TorusGeometry (two tiles, the second one partly off the frame)."""
import collections

import numpy as np
from moog import action_spaces, observers, physics as physics_lib, shapes, tasks
from moog.observers import polygon_modifiers
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators


def get_config(level):
    size, aa, modifier = [((64, 48), 3, None), ((32, 32), 2, polygon_modifiers.TorusGeometry(['movers', 'agent'])),
                          ((128, 128), 2, None), ((50, 37), 1, None), ((30, 22), 3, None),
                          ((150, 41), 1, polygon_modifiers.TorusGeometry(['movers', 'agent']))][level]
    mover_factors = distribs.Product(
        [distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('y', 0.15, 0.85),
         distribs.Discrete('shape', ['triangle', 'star_5', 'circle', 'spoke_4']),
         distribs.Continuous('angle', 0., 2 * np.pi),
         distribs.Continuous('scale', 0.06, 0.2),
         distribs.Continuous('c0', 0., 1.),
         distribs.Discrete('opacity', [255, 255, 140, 70]),
         distribs.Continuous('x_vel', -0.03, 0.03), distribs.Continuous('y_vel', -0.03, 0.03),
         distribs.Continuous('angle_vel', -0.1, 0.1)],
        c1=0.8, c2=0.9)
    agent_factors = distribs.Product(
        [distribs.Continuous('x', 0.3, 0.7), distribs.Continuous('y', 0.3, 0.7)],
        shape='square', scale=0.07, c0=0.33, c1=1., c2=0.7)
    walls = shapes.border_walls(visible_thickness=0.04, c0=0., c1=0., c2=0.4)
    make_movers = sprite_generators.generate_sprites(mover_factors, num_sprites=6)
    make_agent = sprite_generators.generate_sprites(agent_factors, num_sprites=1)

    def state_initializer():
        movers = make_movers()
        agent = make_agent()
        return collections.OrderedDict([('walls', walls), ('movers', movers), ('agent', agent)])

    bounce = physics_lib.Collision(elasticity=1., symmetric=False, update_angle_vel=True)
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.1), 'agent'),
        (bounce, ['movers', 'agent'], 'walls'),
        updates_per_env_step=3)
    task = tasks.CompositeTask(tasks.StayAlive(reward_period=5, reward_value=1.), timeout_steps=25)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(
            image_size=size, anti_aliasing=aa, color_to_rgb='hsv_to_rgb', bg_color=(30, 20, 60),
            polygon_modifier=modifier)},
    }
