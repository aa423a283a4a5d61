"""Base class of environment wrappers (reference: moog/env_wrappers/abstract_wrapper.py:10-80): the wrapped
environment's interface, forwarded."""


class AbstractEnvironmentWrapper(object):
    def __init__(self, environment):
        self._environment = environment

    def reset(self):
        return self._environment.reset()

    def step(self, action):
        return self._environment.step(action)

    def observation(self):
        return self._environment.observation()

    def observation_spec(self):
        return self._environment.observation_spec()

    def action_spec(self):
        return self._environment.action_spec()

    def __getattr__(self, attr):
        # state, meta_state, state_initializer, physics, task, action_space, observers, game_rules, step_count,
        # reset_next_step: whatever the wrapped environment has
        if attr == '_environment':
            raise AttributeError(attr)
        return getattr(self._environment, attr)
