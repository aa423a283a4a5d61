"""Named shapes, wall / grid helpers and circle outlines (reference: moog/shapes.py:11-188)."""
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


def grid_lines(grid_x=0.4, grid_y=0.4, line_thickness=0.01, buffer_border=0., c0=0, c1=0, c2=0,
               opacity=255):
    """Thin rectangles forming a background grid centred on (0.5, 0.5) (shapes.py:80-148):
    the vertical lines first (left to right), then the horizontal ones."""
    half_across = int(np.floor((0.5 + buffer_border) / grid_x))
    half_up = int(np.floor((0.5 + buffer_border) / grid_y))
    xs = np.linspace(start=0.5 - half_across * grid_x, stop=0.5 + half_across * grid_x,
                     num=1 + 2 * half_across)
    ys = np.linspace(start=0.5 - half_up * grid_y, stop=0.5 + half_up * grid_y, num=1 + 2 * half_up)
    factors = dict(x=0., y=0., c0=c0, c1=c1, c2=c2, opacity=opacity)
    out = []

    def add(min_x, max_x, min_y, max_y):
        out.append(sprite.Sprite(shape=np.array(
            [[min_x, min_y], [max_x, min_y], [max_x, max_y], [min_x, max_y]]), **factors))
    for x in xs:
        add(x - 0.5 * line_thickness, x + 0.5 * line_thickness, -1 * buffer_border, 1. + buffer_border)
    for y in ys:
        add(-1 * buffer_border, 1. + buffer_border, y - 0.5 * line_thickness, y + 0.5 * line_thickness)
    return out


def circle_vertices(radius, num_sides=50):
    """shapes.py:151-167"""
    min_theta = 2 * np.pi / num_sides
    thetas = np.linspace(min_theta, 2 * np.pi, num_sides)
    circle = np.stack([np.sin(thetas), np.cos(thetas)], axis=1)
    circle *= radius
    return circle


def annulus_vertices(inner_radius, outer_radius, num_sides=50):
    """shapes.py:170-188 (inner circle, then the outer circle backwards)."""
    inner = circle_vertices(inner_radius, num_sides=num_sides)
    inner = np.concatenate((inner, [inner[0]]), axis=0)
    outer = circle_vertices(outer_radius, num_sides=num_sides)
    outer = np.concatenate((outer, [outer[0]]), axis=0)
    return np.concatenate((inner, outer[::-1]), axis=0)
