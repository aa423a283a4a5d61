"""Colour maps (reference: moog/observers/color_maps.py:21-23).  The device
rasteriser evaluates the same HSV formula per draw; this host copy exists for
API parity (`color_to_rgb=color_maps.hsv_to_rgb`)."""
import colorsys

import numpy as np


def hsv_to_rgb(c):
    return tuple((255 * np.array(colorsys.hsv_to_rgb(*c))).astype(np.uint8))
