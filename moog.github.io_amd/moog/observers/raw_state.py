"""RawState observer (reference: moog/observers/raw_state.py:6-22): the observation is the state
itself.  In the batched engine the state lives in the device records; the observation entry
is a lazy host view: `obs['state'](env=0)` (or `.sprites(env)`) materialises one env's
OrderedDict of sprite attribute dicts, `obs['state'].f64 / .i32` are the record tensors."""


class RawState(object):
    def observation_spec(self):
        raise NotImplementedError


class StateView(object):
    """What a RawState observer returns from the batched engine."""

    def __init__(self, environment):
        self._environment = environment
        self.f64, self.i32 = environment.state_f64, environment.state_i32

    def sprites(self, env=0):
        return self._environment.sprites(env)

    __call__ = sprites
