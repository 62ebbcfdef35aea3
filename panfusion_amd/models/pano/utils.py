"""EPA geometry entry points with the reference's names and return layouts
(reference ``models/pano/utils.py:10-106``).

The denoiser itself never materialises these dense tensors: it consumes the
sparse-flagged ``bias + 1`` tables and PE tables straight from
``engine.EPATables``.  ``get_masks`` / ``get_coords`` exist so that callers of
the reference API (and the parity tests) get the same tensors in the same
shapes; they are thin re-layouts of what the HIP table builder produces.
"""
import torch

from ... import ops


def _flat_cams(cameras):
    host = lambda v: v.detach().cpu().tolist() if isinstance(v, torch.Tensor) else list(v)
    return host(cameras["FoV"]), host(cameras["theta"]), host(cameras["phi"])


def get_masks(pers_h, pers_w, equi_h, equi_w, cameras, device, dtype=torch.float32):
    """-> pers_masks (m, eh, ew, ph, pw), equi_masks (m, ph, pw, eh, ew), values in [-1, 1]
    (reference utils.py:10-84).  ``cameras`` holds m flattened cameras."""
    fov, theta, phi = _flat_cams(cameras)
    m = len(fov)
    bias_e, bias_p, _, _ = ops.epa_tables(fov, theta, phi, pers_h, pers_w, equi_h, equi_w, device)
    E, P = equi_h * equi_w, pers_h * pers_w
    pers_masks = (bias_e.view(E, m, P).permute(1, 0, 2) - 1.0).reshape(m, equi_h, equi_w, pers_h, pers_w)
    equi_masks = (bias_p - 1.0).view(m, pers_h, pers_w, equi_h, equi_w)
    return pers_masks.to(dtype), equi_masks.to(dtype)


def get_coords(pers_h, pers_w, equi_h, equi_w, cameras, device, dtype=torch.float32):
    """-> pers_coords (m, ph, pw, 2) = (lon, lat), equi_coords (eh, ew, 2)
    (reference utils.py:87-106)."""
    fov, theta, phi = _flat_cams(cameras)
    _, _, lonlat = ops.e2p_grid(fov, theta, phi, equi_h, equi_w, pers_h, pers_w, device, want_lonlat=True)
    return lonlat.to(dtype), ops.equi_coords(equi_h, equi_w, device).to(dtype)
