"""Coverage recipe (not a reference task): `generate_sprites(..., fail_gracefully=True)` -- a generator that gives up
after `max_recursion_depth` rejections returns the sprites it has made so far instead of raising RecursionError
(reference moog/state_initialization/sprite_generators.py:24-105, used by red_green.py:128,138 and
bounce_box_contact_prediction.py).  Pinned by golden vectors captured from the reference (tests/golden/sampler_zoo_*.npz).

level 0: six disjoint large squares are asked for at every reset where three to five fit;
level 1: the same generator behind a CreateSprites rule that keeps appending to a crowded layer;
level 2: `shuffle(chain_generators(...))` (sprite_generators.py:108-183): two squares and two circles in a random
         z-order (they overlap, so the order shows in the frames), behind a fixed sprite of the same layer;
level 3: `sample_generator([...], p=[0.3, 0.7])` (sprite_generators.py:131-154): either two disjoint squares or two
         circles that avoid the walls, then an agent placed clear of whichever came out; and a second, uniform choice
         between a star and a triangle.
"""
import collections

from moog import action_spaces, game_rules, observers, physics as physics_lib, shapes, sprite, tasks
from moog.state_initialization import distributions as distribs
from moog.state_initialization import sprite_generators

LAYER_CAPACITY = {'blocks': 8}


def get_config(level):
    block_factors = distribs.Product(
        [distribs.Continuous('x', 0.15, 0.85), distribs.Continuous('y', 0.25, 0.85),
         distribs.Discrete('shape', ['square', 'circle'])],
        scale=0.27, c0=0.55, c1=0.8, c2=0.9)
    crowd = sprite_generators.generate_sprites(block_factors, num_sprites=6, max_recursion_depth=5,
                                               fail_gracefully=True)
    one_more = sprite_generators.generate_sprites(block_factors, num_sprites=2, max_recursion_depth=3,
                                                  fail_gracefully=True)

    def small(shape, hue):
        return sprite_generators.generate_sprites(distribs.Product(
            [distribs.Continuous('x', 0.35, 0.65), distribs.Continuous('y', 0.4, 0.7)],
            shape=shape, scale=0.22, c0=hue, c1=0.9, c2=0.9), num_sprites=2)
    stacked = sprite_generators.shuffle(sprite_generators.chain_generators(small('square', 0.1), small('circle', 0.7)))

    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.05, c0=0., c1=0., c2=0.5)
        if level == 3:
            def roomy(shape, hue):
                return sprite_generators.generate_sprites(distribs.Product(
                    [distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.4, 0.8)],
                    shape=shape, scale=0.15, c0=hue, c1=0.9, c2=0.9), num_sprites=2)
            either = sprite_generators.sample_generator([roomy('square', 0.1), roomy('circle', 0.7)], p=[0.3, 0.7])
            blocks = either(disjoint=True, without_overlapping=walls)
            token = sprite_generators.sample_generator([
                sprite_generators.generate_sprites(distribs.Product(
                    [distribs.Continuous('x', 0.2, 0.8)], y=0.2, shape=shp, scale=0.08, c0=0.9, c1=1., c2=1.))
                for shp in ('star_5', 'triangle')])
            blocks = blocks + token(without_overlapping=walls)
            placed = sprite_generators.generate_sprites(distribs.Product(
                [distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.3, 0.8)],
                shape='circle', scale=0.06, c0=0.33, c1=1., c2=0.7), num_sprites=1)
            agent_list = placed(without_overlapping=walls + blocks)
            return collections.OrderedDict([('walls', walls), ('blocks', blocks), ('agent', agent_list)])
        if level == 2:
            base = sprite.Sprite(x=0.5, y=0.55, shape='hexagon', scale=0.3, c0=0.3, c1=0.5, c2=0.6)
            blocks = [base] + stacked(without_overlapping=walls)
        else:
            blocks = crowd(disjoint=True, without_overlapping=walls) if level == 0 else one_more(without_overlapping=walls)
        agent = sprite.Sprite(x=0.5, y=0.12, shape='circle', scale=0.06, c0=0.33, c1=1., c2=0.7)
        return collections.OrderedDict([('walls', walls), ('blocks', blocks), ('agent', [agent])])

    rules = ()
    if level == 1:
        rules = (game_rules.TimedRule(step_interval=(1, 30), rules=(
            game_rules.CreateSprites('blocks', one_more, without_overlapping=('walls', 'blocks', 'agent')),)),)
    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), 'agent'),
        (physics_lib.Collision(elasticity=0.8, symmetric=False, update_angle_vel=False), 'agent',
         ['walls', 'blocks'] if level < 2 else ['walls']),
        updates_per_env_step=4)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': tasks.CompositeTask(tasks.ContactReward(1., layers_0='agent', layers_1='blocks'),
                                    timeout_steps=9 if level in (0, 3) else 24),
        'action_space': action_spaces.Joystick(scaling_factor=0.01, action_layers='agent'),
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), anti_aliasing=1, color_to_rgb='hsv_to_rgb')},
        'game_rules': rules,
    }
