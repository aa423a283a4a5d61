"""Action-space exercise (parity fixture recipe): Composite (composite.py:11-79) of a Joystick
with momentum, a velocity-controlling Grid and a SetPosition with inertia
(set_position.py:13-67) -- the multi-agent layout of cleanup.py:150-157 plus the eye / hand
example of composite.py:17-31.  Level 1 is a lone SetPosition space.
"""
import collections

from moog import action_spaces
from moog import observers
from moog import physics as physics_lib
from moog import shapes
from moog import sprite
from moog import tasks


def get_config(level=0):
    def state_initializer():
        walls = shapes.border_walls(visible_thickness=0.03, c0=0., c1=0., c2=0.5)
        return collections.OrderedDict([
            ('walls', walls),
            ('prey', [sprite.Sprite(x=0.5, y=0.8, shape='star_5', scale=0.1, c0=0.15, c1=1., c2=1.)]),
            ('agent_0', [sprite.Sprite(x=0.3, y=0.3, shape='circle', scale=0.1, c0=0.2, c1=1., c2=0.7)]),
            ('agent_1', [sprite.Sprite(x=0.7, y=0.3, shape='square', scale=0.1, c0=0.5, c1=1., c2=0.7, mass=2.)]),
            ('eye', [sprite.Sprite(x=0.5, y=0.5, shape='spoke_4', scale=0.05, c0=0., c1=0., c2=1.)]),
        ])

    physics = physics_lib.Physics(
        (physics_lib.Drag(coeff_friction=0.25), ['agent_0', 'agent_1']),
        (physics_lib.Collision(elasticity=0.25, symmetric=False), ['agent_0', 'agent_1'], 'walls'),
        updates_per_env_step=5)
    if level == 0:
        action_space = action_spaces.Composite(
            agent_0=action_spaces.Joystick(scaling_factor=0.01, action_layers='agent_0', momentum=0.5),
            agent_1=action_spaces.Grid(scaling_factor=0.02, action_layers='agent_1', control_velocity=True),
            eye=action_spaces.SetPosition(action_layers='eye', inertia=0.5),
        )
    else:
        action_space = action_spaces.SetPosition(action_layers=('eye', 'agent_0'), inertia=0.)
    task = tasks.CompositeTask(
        tasks.ContactReward(1., layers_0=['agent_0', 'agent_1', 'eye'], layers_1='prey'),
        timeout_steps=15)
    return {
        'state_initializer': state_initializer,
        'physics': physics,
        'task': task,
        'action_space': action_space,
        'observers': {'image': observers.PILRenderer(image_size=(64, 64), color_to_rgb='hsv_to_rgb')},
    }
