"""Restatements of the un-vendored third-party functions the hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  None of these packages is
installed in the build image or present under /root/reference, so their
arithmetic is restated here from the behaviour of the versions the reference
pins (environment_strict.yaml): kornia==0.7.2 (:99), opencv-python==4.9.0.80
(:129), xformers==0.0.22 (:228), torch==2.0.1 (:203).  PARITY UNPINNED for
this file: there is no copy of those packages to check against; the
reference call sites that consume each function are cited.

The same functions back the import shims in oracle/ref_import.py (so that the
reference's own modules run here) and the numpy/torch restatement in
oracle/geometry.py, i.e. both sides of the "oracle vs imported reference"
pinning tests share exactly this file and nothing else.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# cv2.Rodrigues  (reference call sites: external/Perspective_and_Equirectangular/
# e2p.py:25-26, p2e.py:25-26)
# ----------------------------------------------------------------------------
def rodrigues(rvec):
    """Rotation vector -> 3x3 matrix, OpenCV semantics.

    OpenCV computes in double from the input values and converts the result to
    the input depth (float32 in -> float32 out).  R = c*I + (1-c)*r r^T + s*[r]x
    with r the unit axis, evaluated in that order.
    Returns (R, None) like cv2 (the jacobian is never used by the reference).
    """
    rvec = np.asarray(rvec)
    out_dtype = rvec.dtype if rvec.dtype in (np.float32, np.float64) else np.float64
    r = rvec.astype(np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        R = np.eye(3, dtype=np.float64)
    else:
        c = np.cos(theta)
        s = np.sin(theta)
        c1 = 1.0 - c
        itheta = 1.0 / theta
        x, y, z = r[0] * itheta, r[1] * itheta, r[2] * itheta
        rrt = np.array([[x * x, x * y, x * z],
                        [x * y, y * y, y * z],
                        [x * z, y * z, z * z]], dtype=np.float64)
        r_x = np.array([[0.0, -z, y],
                        [z, 0.0, -x],
                        [-y, x, 0.0]], dtype=np.float64)
        R = (c * np.eye(3) + c1 * rrt) + s * r_x
    return R.astype(out_dtype), None


# ----------------------------------------------------------------------------
# kornia.utils.create_meshgrid  (models/pano/utils.py:22-23)
# ----------------------------------------------------------------------------
def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=None):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    base = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)  # W,H,2
    return base.permute(1, 0, 2).unsqueeze(0)  # 1,H,W,2 ; [...,0] = x


# ----------------------------------------------------------------------------
# kornia.geometry.transform.remap  (e2p.py:76, p2e.py:70)
# ----------------------------------------------------------------------------
def remap(image, map_x, map_y, mode="bilinear", padding_mode="zeros",
          align_corners=None, normalized_coordinates=False):
    """grid_sample on pixel maps.  kornia normalises with a precomputed
    factor: ``factor = 2/(size-1)`` (in the map dtype) then
    ``factor * coord - 1`` -- the order matters for the fp32 round trip that
    decides nearest-neighbour indices."""
    b, _, h, w = image.shape
    grid = torch.stack([map_x, map_y], dim=-1).to(image.dtype)
    if not normalized_coordinates:
        hw = torch.stack([torch.tensor(w, dtype=grid.dtype, device=grid.device),
                          torch.tensor(h, dtype=grid.dtype, device=grid.device)])
        factor = torch.tensor(2.0, dtype=grid.dtype, device=grid.device) / (hw - 1).clamp(1e-8)
        grid = factor * grid - 1
    if grid.shape[0] != b:
        grid = grid.expand(b, -1, -1, -1)
    return F.grid_sample(image, grid, mode=mode, padding_mode=padding_mode,
                         align_corners=align_corners)


# ----------------------------------------------------------------------------
# kornia.filters.gaussian_blur2d  (models/pano/utils.py:65,67)
# ----------------------------------------------------------------------------
def gaussian_kernel1d(ksize, sigma, dtype=torch.float32):
    x = torch.arange(ksize, dtype=dtype) - ksize // 2
    if ksize % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * torch.tensor(float(sigma), dtype=dtype).pow(2.0)))
    return g / g.sum()


def gaussian_blur2d(inp, kernel_size, sigma, border_type="reflect", separable=True):
    """Separable Gaussian: horizontal pass (padded in x with border_type),
    then vertical pass (padded in y), each a grouped conv2d."""
    ky, kx = kernel_size
    sy, sx = sigma
    b, c, h, w = inp.shape
    kxv = gaussian_kernel1d(kx, sx, inp.dtype).to(inp.device)
    kyv = gaussian_kernel1d(ky, sy, inp.dtype).to(inp.device)
    x = F.pad(inp, [kx // 2, kx // 2, 0, 0], mode=border_type)
    x = F.conv2d(x.reshape(b * c, 1, h, w + 2 * (kx // 2)), kxv.view(1, 1, 1, kx))
    x = F.pad(x, [0, 0, ky // 2, ky // 2], mode=border_type)
    x = F.conv2d(x, kyv.view(1, 1, ky, 1))
    return x.reshape(b, c, h, w)


# ----------------------------------------------------------------------------
# xformers.ops.memory_efficient_attention  (models/modules/transformer.py:71)
# ----------------------------------------------------------------------------
def memory_efficient_attention(q, k, v, attn_bias=None, p=0.0, scale=None):
    """softmax(q k^T * d^-1/2 + bias) v for 3-D (B*H, N, d) inputs."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    s = torch.einsum("bid,bjd->bij", q, k) * scale
    if attn_bias is not None:
        s = s + attn_bias
    return torch.einsum("bij,bjd->bid", s.softmax(dim=-1), v)
