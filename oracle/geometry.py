"""CPU restatement of the spherical geometry on PanFusion's denoising path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Pinned against the reference's own code by
tests/test_oracle_vs_reference.py (build container only) and against the
fixtures in tests/golden/ everywhere.  Third-party arithmetic (cv2.Rodrigues,
kornia remap/blur, torch grid_sample) comes from oracle/third_party.py and
torch itself -- see the "parity unpinned" note there.

Numerical environment reproduced: numpy==1.26.4 value-based casting
(environment_strict.yaml:116): the Rodrigues vectors and rotation matrices are
float32, the ray arithmetic float64.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import third_party as tp


# ----------------------------------------------------------------------------
# cameras
# ----------------------------------------------------------------------------
def icosahedron_cameras():
    """Centres of the 20 icosahedron faces as (theta, phi) in RADIANS.
    utils/pano.py:34-71.  Rows of five: top cap, upper belt, lower belt
    (shifted by half a step), bottom cap."""
    r_circ = np.sin(2 * np.pi / 5.0)
    r_in = np.sqrt(3) / 12.0 * (3 + np.sqrt(5))
    r_mid = np.cos(np.pi / 5.0)
    step = 2.0 * np.pi / 5.0
    cap = np.pi / 2 - np.arccos(r_in / r_circ)
    belt = np.pi / 2.0 - np.arccos(r_in / r_circ) - 2 * np.arccos(r_in / r_mid)
    thetas, phis = [], []
    for row, (phi, half) in enumerate(((cap, True), (belt, True), (-belt, False), (-cap, False))):
        for i in range(5):
            thetas.append(-np.pi + (step / 2.0 if half else 0.0) + i * step)
            phis.append(phi)
    return np.array(thetas), np.array(phis)


def horizon_cameras(n):
    """utils/pano.py:28-31."""
    theta = np.linspace(0, 2 * np.pi, n, endpoint=False)
    return theta, np.zeros_like(theta)


def camera_rotations(theta_deg, phi_deg):
    """R1 (yaw about z), R2 (pitch about the yawed y axis) as float32 3x3.
    e2p.py:23-26 / p2e.py:23-26 under numpy 1.26 promotion: the float32 axis
    times the float64 scalar stays float32 (the scalar is cast first)."""
    y_axis = np.array([0.0, 1.0, 0.0], np.float32)
    z_axis = np.array([0.0, 0.0, 1.0], np.float32)
    yaw = np.float32(math.radians(float(theta_deg)))
    pitch = np.float32(math.radians(-float(phi_deg)))
    R1, _ = tp.rodrigues(z_axis * yaw)
    R2, _ = tp.rodrigues(np.dot(R1, y_axis) * pitch)
    return R1, R2


# ----------------------------------------------------------------------------
# perspective pixel -> sphere  (e2p.py:9-36)
# ----------------------------------------------------------------------------
def pers_lonlat(wfov, theta, phi, h, w):
    """lon/lat (radians, float64, (h,w)) seen by each pixel of a pinhole view.
    Note the sign: lat is NEGATIVE towards the top image row (e2p.py:35)."""
    hfov = float(h) / w * wfov
    w_len = np.tan(math.radians(wfov / 2.0))
    h_len = np.tan(math.radians(hfov / 2.0))
    ys = np.linspace(-w_len, w_len, w)          # image x -> ray y
    zs = -np.linspace(-h_len, h_len, h)         # image y -> ray z (top = +h_len)
    ray = np.empty((h, w, 3), np.float64)
    ray[..., 0] = np.ones([h, w], np.float32)
    ray[..., 1] = ys[None, :]
    ray[..., 2] = zs[:, None]
    # same evaluation order as the reference: x^2 + y^2 + z^2, sqrt, divide
    norm = np.sqrt(ray[..., 0] ** 2 + ray[..., 1] ** 2 + ray[..., 2] ** 2)
    ray = ray / norm[..., None]
    R1, R2 = camera_rotations(theta, phi)
    v = ray.reshape(-1, 3).T
    v = np.dot(R2, np.dot(R1, v)).T
    lat = np.arcsin(v[:, 2]).reshape(h, w)
    lon = np.arctan2(v[:, 1], v[:, 0]).reshape(h, w)
    return lon, -lat


def e2p_grid(eh, ew, fov, theta, phi, h, w):
    """Sampling positions (pixel units of the (eh,ew) panorama) for each view
    pixel; e2p.py:39-51.  Returns (map_x, map_y) float64."""
    lon, lat = pers_lonlat(fov, theta, phi, h, w)
    cx = (ew - 1) / 2.0
    cy = (eh - 1) / 2.0
    lon = lon / np.pi * 180
    lat = lat / np.pi * 180
    return lon / 180 * cx + cx, lat / 90 * cy + cy


# ----------------------------------------------------------------------------
# panorama pixel -> perspective image  (p2e.py:9-49)
# ----------------------------------------------------------------------------
def p2e_grid(ph, pw, wfov, theta, phi, h, w):
    """For every pixel of an (h,w) panorama: sampling position in the (ph,pw)
    view (0 where not visible) and the visibility mask.  Note u in [0,pw],
    v in [0,ph] (NOT pw-1): p2e.py:41-44."""
    hfov = float(ph) / pw * wfov
    w_len = np.tan(math.radians(wfov / 2.0))
    h_len = np.tan(math.radians(hfov / 2.0))
    lon_deg, lat_deg = np.meshgrid(np.linspace(-180, 180, w), np.linspace(90, -90, h))
    lon_r, lat_r = np.radians(lon_deg), np.radians(lat_deg)
    d = np.stack((np.cos(lon_r) * np.cos(lat_r),
                  np.sin(lon_r) * np.cos(lat_r),
                  np.sin(lat_r)), axis=2)
    R1, R2 = camera_rotations(theta, phi)
    R1i = np.linalg.inv(R1)
    R2i = np.linalg.inv(R2)
    v = d.reshape(-1, 3).T
    v = np.dot(R1i, np.dot(R2i, v)).T.reshape(h, w, 3)
    front = v[..., 0] > 0
    v = v / v[..., 0:1]
    inside = (-w_len < v[..., 1]) & (v[..., 1] < w_len) & (-h_len < v[..., 2]) & (v[..., 2] < h_len)
    u = np.where(inside, (v[..., 1] + w_len) / 2 / w_len * pw, 0)
    vv = np.where(inside, (-v[..., 2] + h_len) / 2 / h_len * ph, 0)
    return u, vv, inside & front


# ----------------------------------------------------------------------------
# resampling  (kornia.remap -> torch grid_sample; e2p.py:54-76, p2e.py:52-71)
# ----------------------------------------------------------------------------
def _scalar_or_index(v, i):
    if hasattr(v, "__len__"):
        v = v[i]
    if isinstance(v, torch.Tensor):
        v = v.item()
    return v


def _per_sample(b, fov, u, v):
    if all(not hasattr(a, "__len__") for a in (fov, u, v)):
        b = 1
    return [(_scalar_or_index(fov, i), _scalar_or_index(u, i), _scalar_or_index(v, i))
            for i in range(b)]


def e2p(e_img, fov_deg, u_deg, v_deg, out_hw, mode=None):
    """Tensor path of e2p.py:54-76 (mode default 'bilinear', utils.py:5-7)."""
    mode = mode or "bilinear"
    b, _, he, we = e_img.shape
    cams = _per_sample(b, fov_deg, u_deg, v_deg)
    maps = [e2p_grid(he, we, f, u, v, out_hw[0], out_hw[1]) for f, u, v in cams]
    mx = torch.from_numpy(np.stack([m[0] for m in maps])).to(e_img.dtype)
    my = torch.from_numpy(np.stack([m[1] for m in maps])).to(e_img.dtype)
    return tp.remap(e_img, mx, my, align_corners=True, mode=mode)


def p2e(p_img, fov_deg, u_deg, v_deg, out_hw, mode=None):
    """Tensor path of p2e.py:52-71: returns (equi, mask)."""
    mode = mode or "bilinear"
    b, _, hp, wp = p_img.shape
    cams = _per_sample(b, fov_deg, u_deg, v_deg)
    maps = [p2e_grid(hp, wp, f, u, v, out_hw[0], out_hw[1]) for f, u, v in cams]
    mx = torch.from_numpy(np.stack([m[0] for m in maps])).to(p_img.dtype)
    my = torch.from_numpy(np.stack([m[1] for m in maps])).to(p_img.dtype)
    mask = torch.from_numpy(np.stack([m[2][None] for m in maps]))
    return tp.remap(p_img, mx, my, align_corners=True, mode=mode) * mask, mask


def sample_position_f32(coord_f64, size):
    """The fp32 position grid_sample actually uses for a float64 pixel map
    entry: cast to fp32, kornia normalise (factor first), torch un-normalise
    (align_corners=True).  numpy float32 arithmetic, element-wise IEEE."""
    x = np.asarray(coord_f64).astype(np.float32)
    factor = np.float32(2.0) / np.float32(size - 1)
    xn = factor * x - np.float32(1.0)
    return ((xn + np.float32(1.0)) / np.float32(2.0)) * np.float32(size - 1)


def nearest_indices(map_x, map_y, src_h, src_w):
    """Integer gather indices of mode='nearest' (round-half-even on the fp32
    position, zeros outside): -1 where the sample falls outside the source."""
    ix = np.rint(sample_position_f32(map_x, src_w)).astype(np.int64)
    iy = np.rint(sample_position_f32(map_y, src_h)).astype(np.int64)
    ok = (ix >= 0) & (ix < src_w) & (iy >= 0) & (iy < src_h)
    return np.where(ok, iy * src_w + ix, -1)


# ----------------------------------------------------------------------------
# circular padding  (utils/pano.py:74-105)
# ----------------------------------------------------------------------------
def pad_pano(pano, padding):
    if padding <= 0:
        return pano
    if pano.ndim not in (4, 5):
        raise NotImplementedError("pano should be 4 or 5 dim")
    return torch.cat([pano[..., -padding:], pano, pano[..., :padding]], dim=-1)


def unpad_pano(pano_pad, padding):
    if padding <= 0:
        return pano_pad
    return pano_pad[..., padding:-padding]


# ----------------------------------------------------------------------------
# EPA correspondence bias  (models/pano/utils.py:10-84)
# ----------------------------------------------------------------------------
def _flat(cameras):
    return cameras["FoV"], cameras["theta"], cameras["phi"]


def get_masks(pers_h, pers_w, equi_h, equi_w, cameras, device="cpu", dtype=torch.float32):
    """Dense soft attention bias in [-1, 1].

    Returns pers_masks (m, eh, ew, ph, pw): for each panorama pixel the bump in
    view m; equi_masks (m, ph, pw, eh, ew): for each view pixel the bump on the
    panorama.  Steps (utils.py line numbers):
      :18-26  identity images: channel k is the one-hot image of pixel k
      :31-38  warp them: p2e of the view identity, e2p of the pano identity
      :49-56  cross-fill: A = clamp(A + B^T), then B = clamp(B + A_new^T)
      :61-68  5x5 sigma=1 Gaussian, replicate border; pano side wrap-padded by 2
      :69-76  divide by the per-image max (1 if empty), map [0,1] -> [-1,1]
    """
    fov, theta, phi = _flat(cameras)
    m = len(fov)
    P, E = pers_h * pers_w, equi_h * equi_w
    eye_p = torch.eye(P, dtype=dtype).reshape(1, P, pers_h, pers_w).expand(m, -1, -1, -1)
    eye_e = torch.eye(E, dtype=dtype).reshape(1, E, equi_h, equi_w).expand(m, -1, -1, -1)
    on_pano = p2e(eye_p.contiguous(), fov, theta, phi, (equi_h, equi_w))[0]   # m,P,eh,ew
    on_view = e2p(eye_e.contiguous(), fov, theta, phi, (pers_h, pers_w))      # m,E,ph,pw
    on_pano = on_pano.reshape(m, P, E)
    on_view = on_view.reshape(m, E, P)
    on_view = torch.clamp(on_view + on_pano.transpose(1, 2), 0, 1)
    on_pano = torch.clamp(on_pano + on_view.transpose(1, 2), 0, 1)

    pv = tp.gaussian_blur2d(on_view.reshape(m * E, 1, pers_h, pers_w), (5, 5), (1.0, 1.0),
                            border_type="replicate")
    pe = pad_pano(on_pano.reshape(m * P, 1, equi_h, equi_w), 2)
    pe = unpad_pano(tp.gaussian_blur2d(pe, (5, 5), (1.0, 1.0), border_type="replicate"), 2)

    def normalise(x):
        peak = torch.amax(x, dim=(1, 2, 3), keepdim=True)
        peak[peak == 0] = 1.0
        return x / peak * 2 - 1

    pers_masks = normalise(pv).reshape(m, equi_h, equi_w, pers_h, pers_w)
    equi_masks = normalise(pe).reshape(m, pers_h, pers_w, equi_h, equi_w)
    return pers_masks, equi_masks


# ----------------------------------------------------------------------------
# EPA positional coordinates and encoding
# ----------------------------------------------------------------------------
def get_coords(pers_h, pers_w, equi_h, equi_w, cameras, device="cpu", dtype=torch.float32):
    """models/pano/utils.py:87-106.  pers (m,h,w,2)=(lon,lat) with lat negative
    at the top row; equi (H,W,2) with lon=linspace(-pi,pi,W) (both endpoints)
    and lat=+pi/2 at the top row."""
    lon = np.linspace(-np.pi, np.pi, equi_w)
    lat = np.linspace(np.pi / 2, -np.pi / 2, equi_h)
    equi = np.stack(np.broadcast_arrays(lon[None, :], lat[:, None]), axis=-1)
    fov, theta, phi = _flat(cameras)
    pers = []
    for i in range(len(fov)):
        lo, la = pers_lonlat(_scalar_or_index(fov, i), _scalar_or_index(theta, i),
                             _scalar_or_index(phi, i), pers_h, pers_w)
        pers.append(np.stack([lo, la], axis=-1))
    return (torch.tensor(np.stack(pers), dtype=dtype), torch.tensor(equi, dtype=dtype))


def spherical_freq_bands(n_freqs):
    """models/modules/transformer.py:174-182 (logscale=True)."""
    base = 2 if n_freqs <= 80 else 5000 ** (1 / (n_freqs / 2.5))
    return base ** torch.linspace(0, n_freqs - 1, n_freqs)


def spherical_pe(coords, freq_bands):
    """transformer.py:185-201: channels = [sin(lon f), sin(lat f), cos(lon f),
    cos(lat f)], each block n_freqs wide, fp32 throughout."""
    lead = coords.shape[:-1]
    arg = coords.reshape(-1, 2, 1) * freq_bands
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1).reshape(*lead, -1)
