"""Observers (reference: moog/observers/__init__.py:3-6)."""
from . import color_maps
from . import polygon_modifiers
from .pil_renderer import PILRenderer
from .raw_state import RawState


class AbstractObserver(object):
    """abstract_observer.py:7-36: `__call__(state)` and `observation_spec()`.  The engine's observers are parameter
    records the rasteriser / the raw-state reader are configured from (PILRenderer, RawState); a config-local subclass
    has no device form and is refused when the environment is built."""

    def __call__(self, state):
        raise NotImplementedError

    def observation_spec(self):
        raise NotImplementedError

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

_submodules(abstract_observer=('AbstractObserver',))
