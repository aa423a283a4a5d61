"""Game rules (reference: moog/game_rules/__init__.py:3-23).

Lowered to the device: VanishOnContact (vanish.py:63-86), ModifySprites whose
modifier is the torus position wrap (modify_sprites.py:8-52 with
chase_avoid_torus.py:144-149), Portal (portal.py:11-76) and config-local rules
that register a lowering (`register_lowering`), e.g. functional_maze.py's
`Booster` (:23-78), recognised by class name and attributes.
"""
import numpy as np

from .. import _abi


class AbstractRule(object):
    """abstract_rule.py:6-34"""

    def reset(self, state, meta_state):
        pass

    def step(self, state, meta_state):
        raise NotImplementedError


class VanishOnContact(AbstractRule):
    def __init__(self, vanishing_layer, contacting_layer):
        self._layer = vanishing_layer
        self._contacting_layer = contacting_layer


class ModifyMetaState(AbstractRule):
    """modify_meta_state.py:7-24: calls `meta_state_modifier(meta_state)` every step.  The
    meta-state is a host-side Python object, so this rule runs on the host; it never
    touches sprites and is skipped by the device lowering."""

    host_side = True

    def __init__(self, meta_state_modifier):
        self._meta_state_modifier = meta_state_modifier

    def step(self, state, meta_state):
        del state
        self._meta_state_modifier(meta_state)


class _ProbeSprite(object):
    def __init__(self, pos):
        self.position = np.array(pos, dtype=float)


class ModifySprites(AbstractRule):
    def __init__(self, layers, modifier, sample_one=False, filter_fn=None):
        if isinstance(layers, str):
            layers = [layers]
        self._layers = list(layers)
        self._modifier = modifier
        self._sample_one = sample_one
        self._filter_fn = filter_fn

    def classify(self):
        """Recognise the modifier by probing it on sample points."""
        if self._sample_one or self._filter_fn is not None:
            raise NotImplementedError('ModifySprites(sample_one/filter_fn) is not lowered')
        pts = [(1.25, -0.25), (0.5, 0.75), (-3.5, 2.0), (0.999, 1e-3)]
        for p in pts:
            s = _ProbeSprite(p)
            try:
                self._modifier(s)
            except Exception as exc:  # pylint: disable=broad-except
                raise NotImplementedError('ModifySprites modifier not recognised: %r' % (exc,))
            if not np.array_equal(np.asarray(s.position), np.remainder(np.array(p), 1)):
                raise NotImplementedError(
                    'ModifySprites modifier is not the torus wrap position = remainder(position, 1)')
        return _abi.MOOG_RULE_TORUS_WRAP


class Portal(AbstractRule):
    def __init__(self, teleporting_layer, portal_layer):
        self._teleporting_layer = teleporting_layer
        self._portal_layer = portal_layer


# ---- lowering registry for config-local rule classes --------------------------
_LOWERINGS = {}


def register_lowering(class_name, fn):
    """fn(rule, layer_index) -> dict(kind=..., l0=..., l1=..., p0=..., p1=..., p2=...)"""
    _LOWERINGS[class_name] = fn


def lookup_lowering(rule):
    for klass in type(rule).__mro__:
        if klass.__name__ in _LOWERINGS:
            return _LOWERINGS[klass.__name__]
    return None


def _lower_booster(rule, layer_index):
    # functional_maze.py:23-78
    return dict(kind=_abi.MOOG_RULE_BOOSTER,
                l0=layer_index(rule._agent_layer), l1=layer_index(rule._booster_layer),
                p0=float(rule._mass_multiplier), p1=float(rule._c2_multiplier),
                p2=float(rule.boost_duration))


register_lowering('Booster', _lower_booster)
