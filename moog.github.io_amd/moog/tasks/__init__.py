"""Tasks (reference: moog/tasks/__init__.py:3-7): CompositeTask
(composite_task.py:17-42), ContactReward (contact_reward.py:20-102), Reset
(reset.py:19-61), StayAlive (stay_alive.py:9-32).  Parameter records; rewards
and reset bookkeeping are computed in the step kernel.
"""
import inspect

import numpy as np

from .. import _abi


class AbstractTask(object):
    pass


class CompositeTask(AbstractTask):
    def __init__(self, *tasks, timeout_steps=np.inf):
        self._tasks = tasks
        self._timeout_steps = timeout_steps


class ContactReward(AbstractTask):
    def __init__(self, reward_fn, layers_0, layers_1, condition=None,
                 reset_steps_after_contact=np.inf):
        # contact_reward.py:44-63: a number or reward_fn(sprite_0, sprite_1); condition takes (sprite_0, sprite_1) or
        # (sprite_0, sprite_1, meta_state).  What a lowered function may read of the meta-state is what lives on the device:
        # the phase a PhaseSequence publishes (`meta_state['phase'] == 'name'`) and the counts Fixation rules publish
        # (moog/_symbolic.py _SymMetaValue); anything else raises NotImplementedError when the config is compiled.
        self._reward = reward_fn
        if condition is not None:
            n_args = len(inspect.signature(condition).parameters)
            if n_args == 3:
                from .. import _symbolic
                cond3 = condition
                condition = lambda s0, s1: cond3(s0, s1, _symbolic._SymMeta())   # noqa: E731
            elif n_args != 2:
                raise ValueError('ContactReward condition must take (sprite_0, sprite_1) or (sprite_0, sprite_1, meta_state)')
        self._condition = condition
        if not isinstance(layers_0, (list, tuple)):
            layers_0 = [layers_0]
        if not isinstance(layers_1, (list, tuple)):
            layers_1 = [layers_1]
        self._layers_0 = list(layers_0)
        self._layers_1 = list(layers_1)
        self._reset_steps_after_contact = reset_steps_after_contact


class _Probe(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Reset(AbstractTask):
    def __init__(self, condition, reward_fn=None, steps_after_condition=np.inf):
        self._with_meta = len(inspect.signature(condition).parameters.values()) == 2
        self._condition = condition
        self._steps_after_condition = steps_after_condition
        self._reward_fn = reward_fn

    def classify(self, layer_names):
        """Recognise `condition` by probing it on synthetic states.

        Recognised forms: "layer L is empty" (chase_avoid_torus.py:114,
        functional_maze.py:202) and "all sprites of layer L have y < c"
        (pong.py:87).  Returns (cond_kind, layer_name, value).
        """
        cond = self._condition
        from .. import _symbolic
        try:   # all(...) / any(...) over one layer, or a test of its first sprite
            kind, layer, node = _symbolic.trace_state_condition(cond, with_meta=self._with_meta)
            code = {'all': _abi.MOOG_COND_ALL_EXPR, 'any': _abi.MOOG_COND_ANY_EXPR,
                    'first': _abi.MOOG_COND_FIRST_EXPR, 'plain': _abi.MOOG_COND_STATE_EXPR}[kind]
            if layer is None:
                layer = layer_names[0]
            if not (kind == 'all' and node.op == 'lt' and node.args[0].key() == ('attr', 0, 'y')
                    and node.args[1].op == 'const'):
                return code, layer, node
        except NotImplementedError:
            try:   # sprites named by position: state[L][k] attributes, overlaps, metadata (bounce_box_contact_prediction.py:123-137)
                # (evaluated whatever the layers hold: the expression is anchored to slots, not to a layer's first live sprite)
                return _abi.MOOG_COND_STATE_EXPR, layer_names[0], _symbolic.trace_state_fixed(cond, self._with_meta)
            except (NotImplementedError, AttributeError, TypeError):
                pass
            if self._with_meta:
                raise

        def state_with(empty=(), y=None):
            return {l: ([] if l in empty else [_Probe(y=(0.5 if y is None else y), x=0.5)])
                    for l in layer_names}

        try:
            base = bool(cond(state_with()))
            flips = [l for l in layer_names if bool(cond(state_with(empty=(l,)))) != base]
            all_empty = bool(cond(state_with(empty=tuple(layer_names))))
        except Exception as exc:  # pylint: disable=broad-except
            raise NotImplementedError('Reset condition not recognised: %r' % (exc,))
        # y-threshold form: true for very low y, false for very high y
        lo = bool(cond(state_with(y=-1e9)))
        hi = bool(cond(state_with(y=1e9)))
        if lo and not hi:
            layer = None
            for l in layer_names:
                st = {k: [_Probe(y=(-1e9 if k != l else 1e9), x=0.5)] for k in layer_names}
                if not bool(cond(st)):
                    if layer is not None:
                        raise NotImplementedError('Reset condition depends on several layers')
                    layer = l
            a, b = -1e9, 1e9
            for _ in range(200):
                mid = 0.5 * (a + b)
                st = {k: [_Probe(y=(mid if k == layer else -1e9), x=0.5)] for k in layer_names}
                if bool(cond(st)):
                    a = mid
                else:
                    b = mid
            c = b
            if abs(c) < 1e-12:
                c = 0.0
            st = {k: [_Probe(y=(c if k == layer else -1e9), x=0.5)] for k in layer_names}
            if bool(cond(st)):
                raise NotImplementedError('Reset condition is not a strict y < c test')
            st = {k: [_Probe(y=-1e9, x=0.5), _Probe(y=(1e9 if k == layer else -1e9), x=0.5)]
                  for k in layer_names}
            if bool(cond(st)):
                raise NotImplementedError('Reset condition is not all(y < c)')
            return _abi.MOOG_COND_ALL_Y_LT, layer, float(c)
        if (not base) and len(flips) == 1 and all_empty:
            return _abi.MOOG_COND_LAYER_EMPTY, flips[0], 0.0
        raise NotImplementedError(
            'Reset condition not recognised (supported: layer empty, all(y < c))')

    def reward_value(self):
        if self._reward_fn is None:
            return 0.
        try:   # a reward that does not read the state (`lambda _: 1`, multi_tracking_with_feature.py:176)
            value = self._reward_fn(None)
        except Exception:  # pylint: disable=broad-except
            value = None
        if isinstance(value, (int, float, np.integer, np.floating)) and not isinstance(value, bool):
            return float(value)
        return 0.   # (reward_node() carries it)

    def reward_node(self):
        """Expression of a reward_fn that reads the state (sprites named by position), or None for a constant."""
        if self._reward_fn is None:
            return None
        try:
            value = self._reward_fn(None)
            if isinstance(value, (int, float, np.integer, np.floating)) and not isinstance(value, bool):
                return None
        except Exception:  # pylint: disable=broad-except
            pass
        from .. import _symbolic
        try:
            return _symbolic.trace_state_fixed(self._reward_fn)
        except (AttributeError, TypeError) as exc:
            raise NotImplementedError('Reset(reward_fn=...) is not lowered: %s' % (exc,))


class StayAlive(AbstractTask):
    def __init__(self, reward_period, reward_value=1.):
        self._reward_period = reward_period
        self._reward_value = reward_value

def _submodules(**modules):
    """The reference keeps one class per file (`from moog.physics import collisions`, `moog.game_rules.vanish.Vanish`);
    here a package is one file, and those module paths are aliases that hold the same objects."""
    import sys
    import types
    for name, names in modules.items():
        m = types.ModuleType(__name__ + '.' + name)
        m.__doc__ = 'Alias module: the reference\'s moog/%s/%s.py (names defined in %s).' % (__name__.split('.')[-1], name, __name__)
        for n in names:
            setattr(m, n, globals()[n])
        sys.modules[m.__name__] = m
        globals()[name] = m

_submodules(abstract_task=('AbstractTask',), composite_task=('CompositeTask',), contact_reward=('ContactReward',), reset=('Reset',), stay_alive=('StayAlive',))
