"""Panorama helpers on the denoising path (mirror of the reference's
``utils/pano.py:21-105``): camera samplers (host, radians) and the circular
width padding."""
import numpy as np
import torch

from .. import ops


def horizon_sample_camera(n):
    """utils/pano.py:28-31."""
    theta = np.linspace(0, 2 * np.pi, n, endpoint=False)
    return theta, np.zeros_like(theta)


def icosahedron_sample_camera():
    """utils/pano.py:34-71: the 20 face centres, four rings of five."""
    r_circ = np.sin(2 * np.pi / 5.0)
    r_in = np.sqrt(3) / 12.0 * (3 + np.sqrt(5))
    r_mid = np.cos(np.pi / 5.0)
    step = 2.0 * np.pi / 5.0
    cap = np.pi / 2 - np.arccos(r_in / r_circ)
    belt = np.pi / 2.0 - np.arccos(r_in / r_circ) - 2 * np.arccos(r_in / r_mid)
    rings = ((cap, step / 2.0), (belt, step / 2.0), (-belt, 0.0), (-cap, 0.0))
    thetas = [-np.pi + off + i * step for _, off in rings for i in range(5)]
    phis = [phi for phi, _ in rings for _ in range(5)]
    return np.array(thetas), np.array(phis)


def pad_pano(pano, padding):
    """Circular padding of the width axis (utils/pano.py:74-99); 4-D or 5-D NCHW."""
    if padding <= 0:
        return pano
    if pano.ndim not in (4, 5):
        raise NotImplementedError("pano should be 4 or 5 dim")
    return ops.pad_width_rows(pano, padding)


def unpad_pano(pano_pad, padding):
    """utils/pano.py:102-105 (a view, like the reference)."""
    if padding <= 0:
        return pano_pad
    return pano_pad[..., padding:-padding]
