"""Polygon modifiers (reference: moog/observers/polygon_modifiers.py:32-38,67-98)."""


class AbstractPolygonModifier(object):
    pass


class DoNothing(AbstractPolygonModifier):
    pass


class TorusGeometry(AbstractPolygonModifier):
    """3x3 duplication at offsets i, j in {-1, 0, 1}, i outer (:87-96).  As in the
    reference every sprite is duplicated regardless of `wrap_layers`."""

    def __init__(self, wrap_layers):
        if not isinstance(wrap_layers, (list, tuple)):
            wrap_layers = [wrap_layers]
        self._wrap_layers = wrap_layers
