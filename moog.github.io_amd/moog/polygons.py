"""Unit-area polygon generators (reference: moog/polygons.py:10-91).

Each returns an [n, 2] float64 vertex array normalised to area 1, vertex order
and starting angle as in the reference so that shape tables match bit for bit.
"""
import numpy as np


def _polar(r, theta):
    return r * np.array([np.cos(theta), np.sin(theta)])


def polygon(num_sides, theta_0=0.):
    """Regular polygon (polygons.py:10-27)."""
    step = 2 * np.pi / num_sides
    pts = np.array([_polar(1, k * step + theta_0) for k in range(num_sides)])
    area = num_sides * np.sin(step / 2) * np.cos(step / 2)
    return np.array(pts) / np.sqrt(area)


def star(num_sides, point_height=1, theta_0=0.):
    """Regular star (polygons.py:30-52): inner vertex, then point, per side."""
    reach = 1 + point_height
    step = 2 * np.pi / num_sides
    pts = np.empty([2 * num_sides, 2])
    for k in range(num_sides):
        pts[2 * k] = _polar(1, k * step + theta_0)
        pts[2 * k + 1] = _polar(reach, (k + 0.5) * step + theta_0)
    area = reach * num_sides * np.sin(step / 2)
    return np.array(pts) / np.sqrt(area)


def spokes(num_sides, spoke_height=1, theta_0=0.):
    """Rectangular-spoked shape (polygons.py:55-91)."""
    step = 2 * np.pi / num_sides
    pts = np.empty([3 * num_sides, 2])
    arm = _polar(spoke_height, -0.5 * step + theta_0)
    for k in range(num_sides):
        corner = _polar(1, k * step + theta_0)
        pts[3 * k] = arm + corner
        pts[3 * k + 1] = corner
        arm = _polar(spoke_height, (k + 0.5) * step + theta_0)
        pts[3 * k + 2] = arm + corner
    area = num_sides * np.sin(step / 2) * (2 + np.cos(step / 2))
    return np.array(pts) / np.sqrt(area)
