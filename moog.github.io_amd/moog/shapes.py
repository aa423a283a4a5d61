"""Named shapes and wall helpers (reference: moog/shapes.py:11-77)."""
import numpy as np

from . import polygons
from . import sprite

# shapes.py:11-24
SHAPES = {
    'triangle': polygons.polygon(num_sides=3, theta_0=np.pi / 2),
    'square': polygons.polygon(num_sides=4, theta_0=np.pi / 4),
    'pentagon': polygons.polygon(num_sides=5, theta_0=np.pi / 2),
    'hexagon': polygons.polygon(num_sides=6),
    'octagon': polygons.polygon(num_sides=8),
    'circle': polygons.polygon(num_sides=30),
    'star_4': polygons.star(num_sides=4, theta_0=np.pi / 4),
    'star_5': polygons.star(num_sides=5, theta_0=np.pi + np.pi / 10),
    'star_6': polygons.star(num_sides=6),
    'spoke_4': polygons.spokes(num_sides=4, theta_0=np.pi / 4),
    'spoke_5': polygons.spokes(num_sides=5, theta_0=np.pi + np.pi / 10),
    'spoke_6': polygons.spokes(num_sides=6),
}


def border_walls(visible_thickness=0.05, total_thickness=0.5, c0=0, c1=0, c2=0, opacity=255):
    """Four wall sprites framing [0,1]^2 (shapes.py:27-77): bottom, top, left, right."""
    lo = visible_thickness - total_thickness
    bottom = np.array([[0., visible_thickness], [1., visible_thickness], [1., lo], [0., lo]])
    span = 1 + total_thickness - 2 * visible_thickness
    outlines = [
        bottom,
        bottom + np.array([[0., span]]),
        np.flip(bottom, axis=1),
        np.flip(bottom, axis=1) + np.array([[span, 0.]]),
    ]
    return [sprite.Sprite(shape=o, x=0., y=0., c0=c0, c1=c1, c2=c2, opacity=opacity)
            for o in outlines]
