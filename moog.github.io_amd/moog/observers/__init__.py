"""Observers (reference: moog/observers/__init__.py:3-6)."""
from . import color_maps
from . import polygon_modifiers
from .pil_renderer import PILRenderer
from .raw_state import RawState
