"""numpy restatement of the reference's dataset-side view cropping, ``external/py360convert`` (SURVEY.md §8f
row 4; north_star names it: "utils/pano, external/py360convert").

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Follows (paths relative to /root/reference):
  * external/py360convert/e2p.py:6-43          -> e2p
  * external/py360convert/utils.py:67-79        -> xyzpers        :231-243 -> rotation_matrix
  * external/py360convert/utils.py:82-91,105-115 -> xyz2uv, uv2coor
  * external/py360convert/utils.py:126-133       -> sample_equirec (scipy.ndimage.map_coordinates, mode='wrap')
scipy's interpolation is third-party C code (scipy.ndimage, ni_interpolation.c): restated here in numpy from its
documented behaviour -- legacy 'wrap' (period len - 1), double accumulation, integer outputs rounded half up and
clamped -- and PINNED bit for bit against the reference's own module running on scipy in this container
(tests/test_oracle_vs_reference.py) and against fixtures generated from it (tests/golden/py360_e2p.npz,
tools/make_golden_py360.py).
"""
import numpy as np


def rotation_matrix(rad, ax):
    ax = np.array(ax)
    ax = ax / np.sqrt((ax ** 2).sum())
    R = np.diag([np.cos(rad)] * 3)
    R = R + np.outer(ax, ax) * (1.0 - np.cos(rad))
    ax = ax * np.sin(rad)
    return R + np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])


def xyzpers(h_fov, v_fov, u, v, out_hw, in_rot):
    out = np.ones((*out_hw, 3), np.float32)
    x_max, y_max = np.tan(h_fov / 2), np.tan(v_fov / 2)
    x_rng = np.linspace(-x_max, x_max, num=out_hw[1], dtype=np.float32)
    y_rng = np.linspace(-y_max, y_max, num=out_hw[0], dtype=np.float32)
    out[..., :2] = np.stack(np.meshgrid(x_rng, -y_rng), -1)
    Rx = rotation_matrix(v, [1, 0, 0])
    Ry = rotation_matrix(u, [0, 1, 0])
    Ri = rotation_matrix(in_rot, np.array([0, 0, 1.0]).dot(Rx).dot(Ry))
    return out.dot(Rx).dot(Ry).dot(Ri)


def coordinates(h, w, fov_deg, u_deg, v_deg, out_hw, in_rot_deg=0):
    """(coor_x, coor_y) float64 of e2p.py:16-32."""
    h_fov, v_fov = fov_deg[0] * np.pi / 180, fov_deg[1] * np.pi / 180
    xyz = xyzpers(h_fov, v_fov, -u_deg * np.pi / 180, v_deg * np.pi / 180, out_hw, in_rot_deg * np.pi / 180)
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    lon = np.arctan2(x, z)
    lat = np.arctan2(y, np.sqrt(x ** 2 + z ** 2))
    return (lon / (2 * np.pi) + 0.5) * w - 0.5, (-lat / np.pi + 0.5) * h - 0.5


def _wrap_coord(c, n):
    c = c.copy()
    sz = n - 1
    neg = c < 0
    c[neg] += sz * ((-c[neg] / sz).astype(np.int64) + 1)
    big = c > sz
    c[big] -= sz * (c[big] / sz).astype(np.int64)
    return c


def _wrap_index(i, n):
    i = i.copy()
    s2 = n - 1
    neg = i < 0
    i[neg] += s2 * ((-i[neg]) // s2 + 1)
    big = i >= n
    i[big] -= s2 * (i[big] // s2)
    return i


def sample_equirec(e_img, coor_x, coor_y, order):
    """map_coordinates(padded, [coor_y, coor_x], order, mode='wrap') on one channel, output in e_img.dtype."""
    H, W = e_img.shape
    pad_u, pad_d = np.roll(e_img[[0]], W // 2, 1), np.roll(e_img[[-1]], W // 2, 1)
    P = np.concatenate([e_img, pad_d, pad_u], 0).astype(np.float64)
    y, x = _wrap_coord(coor_y, H + 2), _wrap_coord(coor_x, W)
    if order == 0:
        acc = P[_wrap_index(np.floor(y + 0.5).astype(np.int64), H + 2), _wrap_index(np.floor(x + 0.5).astype(np.int64), W)]
    else:
        y0, x0 = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
        fy, fx = y - y0, x - x0
        acc = 0.0
        for ky, wy in enumerate((1 - fy, fy)):
            for kx, wx in enumerate((1 - fx, fx)):
                acc = acc + P[_wrap_index(y0 + ky, H + 2), _wrap_index(x0 + kx, W)] * wy * wx
    if np.issubdtype(e_img.dtype, np.integer):
        info = np.iinfo(e_img.dtype)
        acc = np.where(acc > 0, acc + 0.5, 0.0)
        return np.clip(acc, info.min, info.max).astype(e_img.dtype)
    return acc.astype(e_img.dtype)


def e2p(e_img, fov_deg, u_deg, v_deg, out_hw, in_rot_deg=0, mode="bilinear"):
    assert e_img.ndim in (2, 3)
    h, w = e_img.shape[:2]
    order = {"bilinear": 1, "nearest": 0}[mode]
    cx, cy = coordinates(h, w, fov_deg, u_deg, v_deg, out_hw, in_rot_deg)
    if e_img.ndim == 2:
        return sample_equirec(e_img, cx, cy, order)
    return np.stack([sample_equirec(e_img[..., i], cx, cy, order) for i in range(e_img.shape[2])], -1)
