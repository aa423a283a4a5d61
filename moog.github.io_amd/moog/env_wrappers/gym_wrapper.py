"""OpenAI-Gym surface over an Environment (reference:
moog/env_wrappers/gym_wrapper.py:46-142): old 4-tuple API `step -> (obs, reward,
done, info)`, `reset -> obs`, Dict observation space of uint8 Boxes, Box / Discrete
action space.  `gym` itself is optional: minimal space classes are used when it
is not installed.
"""
import numpy as np

from .. import _dm_env as dm_env

try:  # pragma: no cover
    import gym
    from gym import spaces
    _Base = gym.Env
except ImportError:
    _Base = object

    class _Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape)
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

    class _Discrete(object):
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

    class _Dict(object):
        def __init__(self, spaces_):
            self.spaces = dict(spaces_)

        def __getitem__(self, k):
            return self.spaces[k]

    class spaces(object):  # noqa: N801
        Box, Discrete, Dict = _Box, _Discrete, _Dict


def _spec_to_space(spec):
    """gym_wrapper.py:20-43"""
    if isinstance(spec, dm_env.specs.DiscreteArray):
        return spaces.Discrete(spec.num_values)
    if isinstance(spec, dm_env.specs.BoundedArray):
        return spaces.Box(np.asscalar(spec.minimum) if hasattr(np, 'asscalar') else spec.minimum.item(),
                          spec.maximum.item(), shape=spec.shape, dtype=spec.dtype)
    if isinstance(spec, dm_env.specs.Array):
        if spec.dtype == np.uint8:
            return spaces.Box(0, 255, shape=spec.shape, dtype=spec.dtype)
        return spaces.Box(-np.inf, np.inf, shape=spec.shape, dtype=spec.dtype)
    raise ValueError('unsupported spec %r' % (spec,))


class GymWrapper(_Base):
    """gym_wrapper.py:46-142"""
    metadata = {'render.modes': ['rgb_array']}

    def __init__(self, environment):
        self._env = environment
        self._last_render = None
        self._action_space = None
        self._observation_space = None
        self._env.reset()   # gym_wrapper.py:62

    @property
    def observation_space(self):
        if self._observation_space is None:
            comps = {k: _spec_to_space(v) for k, v in self._env.observation_spec().items()}
            self._observation_space = spaces.Dict(comps)
        return self._observation_space

    @property
    def action_space(self):
        if self._action_space is None:
            self._action_space = _spec_to_space(self._env.action_spec())
        return self._action_space

    def _process_obs(self, obs):
        out = {}
        for k, v in obs.items():
            v = np.asarray(v)
            if v.dtype == np.bool_:
                v = v.astype(np.float32)
            out[k] = v
        if 'image' in out:
            self._last_render = out['image']
        return out

    def step(self, action):
        ts = self._env.step(action)
        obs = self._process_obs(ts.observation)
        reward = ts.reward or 0
        done = ts.last()
        return obs, reward, done, {'discount': ts.discount}

    def reset(self):
        ts = self._env.reset()
        return self._process_obs(ts.observation)

    def render(self, mode='rgb_array'):
        if mode != 'rgb_array':
            raise NotImplementedError('only rgb_array rendering is supported')
        return self._last_render

    def close(self):
        pass
