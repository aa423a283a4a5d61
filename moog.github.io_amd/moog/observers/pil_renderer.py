"""PILRenderer parameter record (reference: moog/observers/pil_renderer.py:37-127).

The painter's-algorithm polygon fill Pillow performs for the reference is done
by the HIP scanline rasteriser (csrc/moog_raster.hip), bit-exact on uint8.
"""
import numpy as np

from .. import _dm_env as dm_env
from . import color_maps
from . import polygon_modifiers


class PILRenderer(object):
    def __init__(self, image_size=(64, 64), anti_aliasing=1, bg_color=None, color_to_rgb=None,
                 polygon_modifier=None):
        if int(anti_aliasing) != anti_aliasing or not 1 <= anti_aliasing <= 16:
            raise NotImplementedError('anti_aliasing must be an integer in 1 .. 16')
        anti_aliasing = int(anti_aliasing)
        self._image_size = tuple(image_size)
        self._anti_aliasing = anti_aliasing
        self._canvas_size = (anti_aliasing * image_size[0], anti_aliasing * image_size[1])
        if polygon_modifier is None:
            polygon_modifier = polygon_modifiers.DoNothing()
        self._polygon_modifier = polygon_modifier
        if color_to_rgb is None:
            self._cmap = 'identity'
        elif color_to_rgb == 'hsv_to_rgb' or color_to_rgb is color_maps.hsv_to_rgb:
            self._cmap = 'hsv'
        elif isinstance(color_to_rgb, str):
            color_to_rgb = getattr(color_maps, color_to_rgb)   # pil_renderer.py:74-75
            self._cmap = 'hsv' if color_to_rgb is color_maps.hsv_to_rgb else 'callable'
        elif callable(color_to_rgb):
            # any Python function of the colour triple (pil_renderer.py:72-76,108): it stays on the host -- the environment
            # evaluates it once per distinct colour triple and hands the rasteriser the results per sprite
            # (moog_engine_set_color_override)
            self._cmap = 'callable'
        else:
            raise TypeError('color_to_rgb must be None, the name of a function of color_maps, or a callable')
        self.color_to_rgb = color_to_rgb
        self._bg_color = (0, 0, 0) if bg_color is None else tuple(bg_color)
        self._observation_spec = dm_env.specs.Array(
            shape=self._image_size + (3,), dtype=np.uint8)

    @property
    def polygon_modifier(self):
        return self._polygon_modifier

    def observation_spec(self):
        return self._observation_spec
