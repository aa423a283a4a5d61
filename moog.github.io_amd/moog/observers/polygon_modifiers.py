"""Polygon modifiers (reference: moog/observers/polygon_modifiers.py:32-98)."""


class AbstractPolygonModifier(object):
    pass


class DoNothing(AbstractPolygonModifier):
    pass


class FirstPersonAgent(AbstractPolygonModifier):
    """Translates every polygon so that the first sprite of `agent_layer` is drawn
    at (0.5, 0.5) (polygon_modifiers.py:41-64)."""

    def __init__(self, agent_layer):
        self._agent_layer = agent_layer


class TorusGeometry(AbstractPolygonModifier):
    """3x3 duplication at offsets i, j in {-1, 0, 1}, i outer (:87-96).  As in the
    reference every sprite is duplicated regardless of `wrap_layers`."""

    def __init__(self, wrap_layers):
        if not isinstance(wrap_layers, (list, tuple)):
            wrap_layers = [wrap_layers]
        self._wrap_layers = wrap_layers
