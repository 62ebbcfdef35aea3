"""Panorama helpers on the denoising path (mirror of the reference's
``utils/pano.py:21-105``): camera samplers (host, radians) and the circular
width padding."""
import numpy as np

from .. import ops


def random_sample_camera(n):
    """utils/pano.py:15-25: n uniformly random viewing directions -- normalised Gaussian triples drawn from numpy's
    global RNG (same draws as the reference under the same np.random.seed), phi = asin(z), theta = atan2(x, y)."""
    xyz = np.random.normal(size=(n, 3))
    xyz = xyz / (np.linalg.norm(xyz, axis=-1, keepdims=True) + 1e-9)
    return np.arctan2(xyz[:, 0], xyz[:, 1]), np.arcsin(xyz[:, 2].clip(-1, 1))


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


class Equirectangular:
    """The dataset's panorama holder (utils/pano.py:142-171), numpy image in ``.equirectangular``; the view crops
    (``to_perspective``, and the batched ``to_perspectives`` the dataset loop wants) run on the HIP kernel behind
    ``external.py360convert.e2p``."""

    def __init__(self, equirectangular):
        self.equirectangular = equirectangular

    def to_perspective(self, fov, yaw, pitch, hw, mode="bilinear"):
        from ..external.py360convert import e2p
        return e2p(self.equirectangular, fov, yaw, pitch, hw, mode=mode)

    def to_perspectives(self, fov, yaws, pitches, hw, mode="bilinear"):
        """All crops of dataset/PanoDataset.py:136-139 in one launch: [m, h, w, C]."""
        from ..external.py360convert import e2p_views
        return e2p_views(self.equirectangular, fov, yaws, pitches, hw, mode=mode)

    def rotate(self, degree):
        if degree % 360 == 0:
            return
        self.equirectangular = np.roll(self.equirectangular, int(degree / 360 * self.equirectangular.shape[1]), axis=1)

    def flip(self, flip=True):
        if flip:
            self.equirectangular = np.flip(self.equirectangular, 1)
