"""Torch-tensor front end of the C-ABI HIP kernels.

PyTorch is used here for device memory and streams only: every function below
hands raw device pointers + sizes to ``libpanfusion_hip.so`` on the caller's
current stream.  No function has a torch / CPU fallback.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import PF_BF16, PF_F16, PF_F32, AttnBwdDesc, AttnDesc, ConvDesc, LinearWsDesc, check

_DT = {torch.bfloat16: PF_BF16, torch.float16: PF_F16, torch.float32: PF_F32}

# When set to a list, every MFMA-kernel launch appends (kernel family, algorithmic FLOPs, start event,
# end event) -- used by bench.py's instrumented step for the roofline line.  None = no overhead.
TRACE = None


class _PlanCache(__import__("threading").local):
    """conv_gemm: problem shape -> [split-K scratch bytes, GroupNorm-moment rows, descriptor] as the library plans it.  Per host
    THREAD (the cached descriptor's pointer members are rewritten on every call: two threads issuing the same shape must not
    share one) and bounded (variable batch sizes / resolutions: the oldest shape goes first)."""
    LIMIT = 4096

    def __init__(self):
        self.plans = {}


_PLANS = _PlanCache()
SPLITK_INKERNEL = os.environ.get("PF_SPLITK_INKERNEL") == "1"    # A/B switch, read once (as the library reads it)


def _traced(name, flops, launch, tag=""):
    if TRACE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    TRACE.append((name, float(flops), e0, e1, tag))


def dt(t):
    return _DT[t.dtype if isinstance(t, torch.Tensor) else t]


try:                                     # raw handle of the current stream without building a torch.cuda.Stream object:
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice      # 0.3 us against 8 us per
except AttributeError:                   # launch -- a third of the host time of a launch-bound step (tools/train_profile_host.py)
    _raw_stream = _cur_device = None


def _stream():
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    """Row stride (elements) of a [.., rows, cols] tensor or column-sliced view of one."""
    assert t.stride(-1) == 1, "innermost dimension must be contiguous"
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


def _cams(fov, theta, phi):
    """Host double arrays (degrees) for the C ABI; accepts tensors / arrays / lists."""
    def host(v):
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    f, t, p = host(fov), host(theta), host(phi)
    n = max(len(f), len(t), len(p))
    f, t, p = (np.ascontiguousarray(np.broadcast_to(a, (n,))) if len(a) != n else a for a in (f, t, p))
    as_p = lambda a: a.ctypes.data_as(_lib.c_dp)
    return n, (f, t, p), (as_p(f), as_p(t), as_p(p))


# ---------------------------------------------------------------------------- geometry
def e2p_grid(fov, theta, phi, eh, ew, h, w, device, want_lonlat=False):
    n, keep, (f, t, p) = _cams(fov, theta, phi)
    mx = torch.empty(n, h, w, device=device, dtype=torch.float32)
    my = torch.empty_like(mx)
    ll = torch.empty(n, h, w, 2, device=device, dtype=torch.float32) if want_lonlat else None
    check(_lib.lib().pf_e2p_grid(f, t, p, n, eh, ew, h, w, _p(mx), _p(my), _p(ll), _stream()), "pf_e2p_grid")
    return (mx, my, ll) if want_lonlat else (mx, my)


def p2e_grid(fov, theta, phi, ph, pw, h, w, device):
    n, keep, (f, t, p) = _cams(fov, theta, phi)
    mu = torch.empty(n, h, w, device=device, dtype=torch.float32)
    mv = torch.empty_like(mu)
    mask = torch.empty(n, h, w, device=device, dtype=torch.uint8)
    check(_lib.lib().pf_p2e_grid(f, t, p, n, ph, pw, h, w, _p(mu), _p(mv), _p(mask), _stream()), "pf_p2e_grid")
    return mu, mv, mask


def nearest_indices(map_x, map_y, src_h, src_w):
    idx = torch.empty(map_x.shape, device=map_x.device, dtype=torch.int32)
    check(_lib.lib().pf_nearest_indices(_p(map_x), _p(map_y), map_x.numel(), src_h, src_w, _p(idx), _stream()),
          "pf_nearest_indices")
    return idx


def remap(src, map_x, map_y, mode, mask=None):
    """src NCHW (fp32 / 16-bit), maps (B or 1, ho, wo) fp32."""
    src = src.contiguous()
    B, Cc, hs, ws = src.shape
    mb, ho, wo = map_x.shape
    out = torch.empty(B, Cc, ho, wo, device=src.device, dtype=src.dtype)
    check(_lib.lib().pf_remap(_p(src), dt(src), B, Cc, hs, ws, _p(map_x), _p(map_y), _p(mask), mb, ho, wo,
                              {"nearest": 0, "bilinear": 1}[mode], _p(out), _stream()), "pf_remap")
    return out


def equi_coords(H, W, device):
    out = torch.empty(H, W, 2, device=device, dtype=torch.float32)
    check(_lib.lib().pf_equi_coords(H, W, _p(out), _stream()), "pf_equi_coords")
    return out


def spherical_pe(coords, freq_bands):
    coords = coords.contiguous()
    n = coords.numel() // 2
    nf = freq_bands.numel()
    out = torch.empty(*coords.shape[:-1], 4 * nf, device=coords.device, dtype=torch.float32)
    check(_lib.lib().pf_spherical_pe(_p(coords), n, _p(freq_bands), nf, _p(out), _stream()), "pf_spherical_pe")
    return out


def epa_tables(fov, theta, phi, ph, pw, eh, ew, device):
    """-> bias_e [E, m*P], bias_p [m*P, E] (fp32, mask + 1), flags_e, flags_p (uint8 tile maps)."""
    m, keep, (f, t, p) = _cams(fov, theta, phi)
    E, P = eh * ew, ph * pw
    mP = m * P
    bias_e = torch.empty(E, mP, device=device, dtype=torch.float32)
    bias_p = torch.empty(mP, E, device=device, dtype=torch.float32)
    flags_e = torch.empty((E + 31) // 32, (mP + 31) // 32, device=device, dtype=torch.uint8)
    flags_p = torch.empty((mP + 31) // 32, (E + 31) // 32, device=device, dtype=torch.uint8)
    nbytes = _lib.lib().pf_epa_tables_workspace_size(m, ph, pw, eh, ew)
    ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
    check(_lib.lib().pf_epa_tables_build(f, t, p, m, ph, pw, eh, ew, _p(bias_e), _p(bias_p), _p(flags_e),
                                         _p(flags_p), _p(ws), nbytes, _stream()), "pf_epa_tables_build")
    return bias_e, bias_p, flags_e, flags_p


# ---------------------------------------------------------------------------- norms / pointwise
GN_FROM_EPILOGUE = os.environ.get("PF_GN_EPILOGUE", "1") != "0"      # A/B: 0 = always the separate statistics pass

# Arrival counters of the in-kernel split-K combine (pf_conv_desc.tickets): one zeroed int32 ring per device, handed out in
# slices of _TICKET_SLICE counters per split launch.  A slice is zero again when its launch has run (the last-arriving
# workgroup resets its counter) and comes round again only _TICKET_SLICES split launches later -- several denoiser steps --
# so no two launches that could overlap in time ever share one.  Allocated outside any graph capture (the warm-up pass).
_TICKET_SLICE, _TICKET_SLICES = 1024, 1024
_TICKETS = {}


def _ticket_slice(device):
    key = (device.type, device.index)
    ring = _TICKETS.get(key)
    if ring is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the split-K counter ring must be allocated before a graph capture (run one eager pass first)")
        ring = _TICKETS[key] = [torch.zeros(_TICKET_SLICE * _TICKET_SLICES, device=device, dtype=torch.int32), 0]
    off = ring[1]
    ring[1] = (off + 1) % _TICKET_SLICES
    return ring[0].data_ptr() + 4 * _TICKET_SLICE * off


def carry(dst, src):
    """dst = a view / reshape of src: hand over the GroupNorm moments a GEMM epilogue attached to src (`_pf_gn`)."""
    st = getattr(src, "_pf_gn", None)
    if st is not None:
        dst._pf_gn = st
    return dst


def groupnorm_scale_shift(x0, x1, n_img, hw, groups, eps, gamma, beta, ws=None, wrap=None):
    """x0 [n, hw, c0] (+ x1 [n, hw, c1] concatenated along channels) -> (scale, shift) [n, C] fp32.
    wrap = (image width, p): statistics of the tensor as if its width had been padded circularly by p columns (pad_pano).
    When every source still carries the per-column moments the GEMM that produced it left behind (conv_gemm(gn_stats=True)
    -> tensor attribute `_pf_gn` = (partials, rows per part)), the statistics come from those -- no pass over the tensors."""
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    Cc = c0 + c1
    scale = torch.empty(n_img, Cc, device=x0.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    st0 = getattr(x0, "_pf_gn", None)
    st1 = getattr(x1, "_pf_gn", None) if x1 is not None else None
    if wrap is not None and wrap[1] > 0:
        nbytes = _lib.lib().pf_groupnorm_workspace_size(n_img, hw, Cc)
        ws = torch.empty(nbytes, device=x0.device, dtype=torch.uint8)
        check(_lib.lib().pf_groupnorm_stats_wrap(_p(x0), c0, _p(x1), c1, dt(x0), n_img, hw // wrap[0], wrap[0], wrap[1], groups, eps,
                                                 _p(gamma), _p(beta), _p(scale), _p(shift), _p(ws), ws.numel(), _stream()),
              "pf_groupnorm_stats_wrap")
        return scale, shift
    usable = st0 is not None and (x1 is None or st1 is not None) and (Cc // groups) % 2 == 0 and c0 % 2 == 0
    if usable and (hw % st0[1] or (st1 is not None and hw % st1[1])):
        usable = False                # (a linear layer's moment runs need not respect this consumer's image boundaries)
    if usable:
        check(_lib.lib().pf_groupnorm_from_partials(_p(st0[0]), c0, st0[1], _p(st1[0]) if st1 else None, c1,
                                                    st1[1] if st1 else st0[1], n_img, hw, groups, eps, _p(gamma), _p(beta),
                                                    _p(scale), _p(shift), _stream()), "pf_groupnorm_from_partials")
        return scale, shift
    nbytes = _lib.lib().pf_groupnorm_workspace_size(n_img, hw, Cc)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, device=x0.device, dtype=torch.uint8)
    check(_lib.lib().pf_groupnorm_stats(_p(x0), c0, _p(x1), c1, dt(x0), n_img, hw, groups, eps, _p(gamma),
                                        _p(beta), _p(scale), _p(shift), _p(ws), ws.numel(), _stream()),
          "pf_groupnorm_stats")
    return scale, shift


def scale_shift_act(x0, x1, n_img, hw, scale, shift, act, out=None, out_dtype=None, split=False, raw_pair=False):
    """y = act(x * scale + shift) over the channel concat (x0 | x1); scale = shift = None: y = act(x).
    Sources 16-bit or fp32.  out_dtype: 16-bit type (default: the sources') or fp32; split=True: the
    16-bit pair [.., hi(C) | lo(C)] (A operand of a split-precision GEMM, engine.exact_gemm).
    raw_pair=True (fp32 sources, plain 16-bit y): also returns the UN-normalised input as that pair, written in the
    same pass -> (y, pair)."""
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    out_dtype = out_dtype or x0.dtype
    if out is None:
        out = torch.empty(n_img, hw, (c0 + c1) * (2 if split else 1), device=x0.device, dtype=out_dtype)
    else:
        _written(out)
    if raw_pair:
        assert x0.dtype == torch.float32 and not split and out_dtype != torch.float32
        pair = torch.empty(n_img, hw, 2 * (c0 + c1), device=x0.device, dtype=out_dtype)
        check(_lib.lib().pf_scale_shift_act_pair(_p(x0), c0, _p(x1), c1, n_img, hw, _p(scale), _p(shift), int(act),
                                                 dt(out_dtype), _p(out), _p(pair), _stream()), "pf_scale_shift_act_pair")
        return out, pair
    check(_lib.lib().pf_scale_shift_act(_p(x0), c0, _p(x1), c1, dt(x0), n_img, hw, _p(scale), _p(shift),
                                        int(act), dt(out_dtype), int(split), _p(out), _stream()), "pf_scale_shift_act")
    return out


def layernorm(x, gamma, beta, eps=1e-5, pe=None, out=None, out_dtype=None):
    """x [rows, C] 16-bit or fp32 -> 16-bit (out_dtype, default x.dtype); pe optional fp32 [pe_rows, C]
    (row r uses pe[r % pe_rows])."""
    rows, Cc = x.shape
    if out is None:
        out = torch.empty(rows, Cc, device=x.device, dtype=out_dtype or x.dtype)
    else:
        _written(out)
    check(_lib.lib().pf_layernorm(_p(x), _p(pe), 0 if pe is None else pe.shape[0], dt(x), rows, Cc,
                                  _p(gamma), _p(beta), eps, dt(out), _p(out), _stream()), "pf_layernorm")
    return out


def geglu(x, out=None):
    rows, two_inner = x.shape
    inner = two_inner // 2
    if out is None:
        out = torch.empty(rows, inner, device=x.device, dtype=x.dtype)
    else:
        _written(out)
    check(_lib.lib().pf_geglu(_p(x), dt(x), rows, inner, _p(out), _stream()), "pf_geglu")
    return out


def timestep_features(t, dim, dtype):
    """t: 1-D int64 (any stride: `timestep[:, 0]` of a (b, m) tensor is read in place)."""
    if t.dtype != torch.int64 or t.dim() != 1:
        t = t.to(torch.int64).reshape(-1).contiguous()
    out = torch.empty(t.numel(), dim, device=t.device, dtype=dtype)
    check(_lib.lib().pf_timestep_features_strided(_p(t), t.stride(0) if t.numel() > 1 else 1, t.numel(), dim, dt(dtype), _p(out), _stream()),
          "pf_timestep_features")
    return out


def _written(out):
    """An op wrote into a caller's `out=` buffer: GroupNorm moments it carried (`_pf_gn`, attached by a GEMM epilogue) described
    its previous contents."""
    if hasattr(out, "_pf_gn"):
        del out._pf_gn
    return out


def silu(x, out=None):
    if out is None:
        out = torch.empty_like(x)
    else:
        _written(out)
    check(_lib.lib().pf_silu(_p(x), dt(x), x.numel(), _p(out), _stream()), "pf_silu")
    return out


def add(a, b, out=None):
    """a + b in a's dtype (b may have another one: fp32 stream + 16-bit ControlNet residual)."""
    if out is None:
        out = torch.empty_like(a)
    else:
        _written(out)
    check(_lib.lib().pf_add(_p(a), dt(a), _p(b), dt(b), a.numel(), _p(out), _stream()), "pf_add")
    return out


def pad_width(x, pad, out=None):
    """x NHWC [n, h, w, C] -> circularly padded [n, h, w + 2 pad, C]."""
    n, h, w, Cc = x.shape
    if out is None:
        out = torch.empty(n, h, w + 2 * pad, Cc, device=x.device, dtype=x.dtype)
    else:
        _written(out)
    check(_lib.lib().pf_pad_width(_p(x), dt(x), n, h, w, Cc, pad, _p(out), _stream()), "pf_pad_width")
    return out


def crop_width(x, crop, out=None):
    n, h, w, Cc = x.shape
    if out is None:
        out = torch.empty(n, h, w - 2 * crop, Cc, device=x.device, dtype=x.dtype)
    else:
        _written(out)
    check(_lib.lib().pf_crop_width(_p(x), dt(x), n, h, w, Cc, crop, _p(out), _stream()), "pf_crop_width")
    return out


def pad_width_rows(x, pad):
    """Circular pad of the last axis of a contiguous tensor of any rank (pad_pano on NCHW)."""
    x = x.contiguous()
    w = x.shape[-1]
    rows = x.numel() // w
    out = torch.empty(*x.shape[:-1], w + 2 * pad, device=x.device, dtype=x.dtype)
    check(_lib.lib().pf_pad_width_rows(_p(x), x.element_size(), rows, w, pad, _p(out), _stream()),
          "pf_pad_width_rows")
    return out


def roll_width(x, shift, out=None):
    """torch.roll(x, shift, dims=-1) on a contiguous tensor."""
    x = x.contiguous()
    w = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    else:
        _written(out)
    check(_lib.lib().pf_roll_width_rows(_p(x), x.element_size(), x.numel() // w, w, int(shift), _p(out), _stream()),
          "pf_roll_width_rows")
    return out


def nchw_to_nhwc(x, dtype, out=None):
    x = x.contiguous()
    n, Cc, h, w = x.shape
    if out is None:
        out = torch.empty(n, h, w, Cc, device=x.device, dtype=dtype)
    else:
        _written(out)
    check(_lib.lib().pf_nchw_to_nhwc(_p(x), dt(x), n, Cc, h, w, dt(dtype), _p(out), _stream()), "pf_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, dtype, out=None):
    n, h, w, Cc = x.shape
    if out is None:
        out = torch.empty(n, Cc, h, w, device=x.device, dtype=dtype)
    else:
        _written(out)
    check(_lib.lib().pf_nhwc_to_nchw(_p(x), dt(x), n, Cc, h, w, dt(dtype), _p(out), _stream()), "pf_nhwc_to_nchw")
    return out


def embed_tokens(ids, tok, pos, Lp, out_dtype):
    """ids int64 [B, L]; tok [vocab, C], pos [>= L, C] fp32 -> [B, Lp, C] (rows >= L are zero)."""
    ids = ids.to(torch.int64).contiguous()
    B, L = ids.shape
    Cc = tok.shape[1]
    out = torch.empty(B, Lp, Cc, device=tok.device, dtype=out_dtype)
    check(_lib.lib().pf_embed_tokens(_p(ids), B, L, Lp, Cc, tok.shape[0], _p(tok), _p(pos), dt(out_dtype), _p(out), _stream()),
          "pf_embed_tokens")
    return out


def softmax_rows(scores, scale, out_dtype, out=None):
    """scores fp32 [..., rows, n] (contiguous rows) -> probabilities [..., rows, n] in out_dtype (16-bit)."""
    n = scores.shape[-1]
    rows = scores.numel() // n
    if out is None:
        out = torch.empty(scores.shape, device=scores.device, dtype=out_dtype)
    check(_lib.lib().pf_softmax_rows(_p(scores), rows, n, _ld(scores), float(scale), dt(out), _p(out), _ld(out), _stream()),
          "pf_softmax_rows")
    return out


def axpby(x, y, a, b, out=None):
    """a * x + b * y on fp32 tensors of one shape."""
    x, y = x.float().contiguous(), y.float().contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().pf_axpby(_p(x), _p(y), float(a), float(b), x.numel(), _p(out), _stream()), "pf_axpby")
    return out


def vae_sample(moments, eps, scale):
    """moments fp32 NHWC [n, h, w, 2L] = (mean | logvar), eps fp32 NCHW [n, L, h, w] -> z fp32 NCHW
    = (mean + exp(0.5 clamp(logvar, -30, 20)) eps) * scale."""
    n, h, w, L2 = moments.shape
    z = torch.empty(n, L2 // 2, h, w, device=moments.device, dtype=torch.float32)
    check(_lib.lib().pf_vae_sample(_p(moments), _p(eps), n, L2 // 2, h * w, float(scale), _p(z), _stream()), "pf_vae_sample")
    return z


def tensor_to_image(x):
    """fp32 NCHW image in [-1, 1] -> uint8 NHWC (models/modules/utils.py:9-15)."""
    x = x.float().contiguous()
    n, Cc, h, w = x.shape
    out = torch.empty(n, h, w, Cc, device=x.device, dtype=torch.uint8)
    check(_lib.lib().pf_tensor_to_image(_p(x), n, Cc, h, w, _p(out), _stream()), "pf_tensor_to_image")
    return out


def cfg_ddim_step(x, eps_uncond, eps_cond, guidance, coef, roll=0, out=None):
    """x / eps fp32 [..., W]; coef = (sqrt a_t, sqrt(1-a_t), sqrt a_prev, sqrt(1-a_prev))."""
    W = x.shape[-1]
    rows = x.numel() // W
    if out is None:
        out = torch.empty_like(x)
    sa, sb, sap, sbp = (float(c) for c in coef)
    check(_lib.lib().pf_cfg_ddim_step(_p(x), _p(eps_uncond), _p(eps_cond), float(guidance), sa, sb, sap, sbp,
                                      rows, W, int(roll), _p(out), _stream()), "pf_cfg_ddim_step")
    return out


def cfg_ddim_step_pair(x, eps_uncond, eps_cond, guidance, coef, roll=0, out=None, out2=None, tstep=None, t_next=0):
    """cfg_ddim_step as the loop's one-launch state update: ``out`` may be ``x`` itself for any roll, ``out2`` receives a second
    copy (the other half of the CFG pair the next denoiser call reads), ``tstep`` (int64 device tensor) is set to ``t_next``."""
    W = x.shape[-1]
    rows = x.numel() // W
    if out is None:
        out = torch.empty_like(x)
    assert x.is_contiguous() and out.is_contiguous() and (out2 is None or out2.is_contiguous())
    assert tstep is None or (tstep.dtype == torch.int64 and tstep.is_contiguous())
    sa, sb, sap, sbp = (float(c) for c in coef)
    check(_lib.lib().pf_cfg_ddim_step_pair(_p(x), _p(eps_uncond), _p(eps_cond), float(guidance), sa, sb, sap, sbp,
                                           rows, W, int(roll), _p(out), _p(out2) if out2 is not None else None,
                                           _p(tstep) if tstep is not None else None, tstep.numel() if tstep is not None else 0,
                                           int(t_next), _stream()), "pf_cfg_ddim_step_pair")
    return out


# ---------------------------------------------------------------------------- GEMM / conv
def conv_gemm(a0, w, n_out, *, a1=None, n_img=1, h_in=1, w_in=None, ksize=1, stride=1, pad=0, upsample=0,
              bias=None, rowvec=None, residual=None, out=None, out_dtype=None, batch=1,
              a_bstride=0, w_bstride=0, out_bstride=0, res_bstride=0, a0_ld=None, a1_ld=None, c0=None, c1=None,
              out_ld=None, res_ld=None, geglu=False, algo_k=None, split_out=False, pad_hi=0, gn_stats=False,
              wrap_pad=0, crop=0, split3=False, plan_only=False, subpixel=False):
    """out[m, n] = sum_k A[m, k] W[n, k] (+bias +rowvec[img] +residual).  a0/a1 NHWC, the last
    dim is the channel stride; returns [M, n_out] (M = n_img * h_out * w_out).  The output takes the
    residual's dtype unless out_dtype says otherwise (fp32 residual stream in, fp32 out).
    algo_k: K of the layer for the FLOP count when the launched K carries split-precision passes.
    split_out: the result leaves as the 16-bit pair [M, 2 n_out] (per 32 columns [hi | lo]: operand of engine.exact_gemm).
    split3: a0 IS such a pair (c0 = 2 x channels) and w the matching [W_hi | W_lo] packing: three products per K block.
    pad_hi = 1: one more zero row / column at the bottom / right (F.pad(x, (0, 1, 0, 1)) of the VAE encoder's down-convs).
    gn_stats: the result feeds a GroupNorm -- where the kernel serving this problem can, its epilogue leaves the per-column
    moments of the output behind (attribute `_pf_gn` of the returned tensor, read by groupnorm_scale_shift).
    wrap_pad / crop: the input is read as if its width had been padded circularly by wrap_pad columns (pad_pano), the output
    loses `crop` columns on both sides (unpad_pano): pad -> conv -> crop of the panorama branch without the padded copies.
    subpixel: an upsampling convolution (ksize 3, upsample 1) as four 2x2 phase convolutions on the low-resolution grid; w = engine._subpixel_weight
    ([4 * n_out, 4 * C]): 4 instead of 9 MACs per output value and input channel (pf_conv_desc.subpixel).
    plan_only: launch nothing, return pf_conv_gemm_kernel_id of the problem (0 / 1: the 16x16x32 tile kernels, 2: the 32x32x16 kernel)."""
    c0 = c0 if c0 is not None else a0.shape[-1]
    c1 = (c1 if c1 is not None else a1.shape[-1]) if a1 is not None else 0
    if w_in is None:
        w_in = a0.numel() // (a0.shape[-1] * max(batch, 1)) if ksize == 1 else None
    hl, wl = h_in << upsample, (w_in + 2 * wrap_pad) << upsample
    h_out = (hl + 2 * pad + pad_hi - ksize) // stride + 1
    w_out = (wl + 2 * pad + pad_hi - ksize) // stride + 1 - 2 * crop
    M = n_img * h_out * w_out
    if out is not None:
        out_dtype = out.dtype
    if split_out:
        out_dtype = a0.dtype
    out_dtype = out_dtype or (residual.dtype if residual is not None else a0.dtype)
    n_store = n_out // 2 if geglu else (2 * n_out if split_out else n_out)
    if out is None:
        out = torch.empty((batch, M, n_store) if batch > 1 else (M, n_store), device=a0.device, dtype=out_dtype)
    a0_ld = a0_ld if a0_ld is not None else _ld(a0)
    a1_ld = (a1_ld if a1_ld is not None else _ld(a1)) if a1 is not None else 0
    rowvec_ld = _ld(rowvec) if rowvec is not None else 0
    res_ld = (res_ld if res_ld is not None else _ld(residual)) if residual is not None else 0
    res_dtype = dt(residual) if residual is not None else dt(a0)
    out_ld = out_ld if out_ld is not None else _ld(out)
    epilogue = 1 if geglu else (2 if split_out else 0)
    # One descriptor per problem SHAPE, filled once and reused (a launch-bound training step issues ~1500 of these per step:
    # the ~35 scalar members cost more host time than the call); with it the library's plan for the shape (split-K scratch,
    # GroupNorm-moment rows), asked once.  Per call only the pointers change.
    pkey = (c0, c1, a0_ld, a1_ld, n_img, h_in, w_in, h_out, w_out, ksize, stride, pad, upsample, n_out, batch, epilogue, wrap_pad,
            crop, dt(a0), dt(out_dtype), residual is not None, res_dtype, res_ld, rowvec is not None, rowvec_ld, bias is not None,
            out_ld, a_bstride, w_bstride, out_bstride, res_bstride, bool(split3), bool(subpixel))
    plans = _PLANS.plans
    plan = plans.get(pkey)
    if plan is None:
        d = ConvDesc()
        d.c0, d.c1, d.a0_ld, d.a1_ld = c0, c1, a0_ld, a1_ld
        d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = n_img, h_in, w_in, h_out, w_out
        d.ksize, d.stride, d.pad, d.upsample = ksize, stride, pad, upsample
        d.n_out, d.rowvec_ld, d.res_ld, d.res_dtype, d.out_ld = n_out, rowvec_ld, res_ld, res_dtype, out_ld
        d.out_dtype, d.dtype = dt(out_dtype), dt(a0)
        d.batch = batch
        d.a_bstride, d.w_bstride, d.out_bstride, d.res_bstride = a_bstride, w_bstride, out_bstride, res_bstride
        d.epilogue = epilogue
        d.wrap_pad, d.crop = wrap_pad, crop
        d.split3 = int(bool(split3))
        d.subpixel = int(bool(subpixel))
        d.a0, d.a1, d.w, d.bias, d.rowvec, d.residual, d.out = _p(a0), _p(a1), _p(w), _p(bias), _p(rowvec), _p(residual), _p(out)
        if len(plans) >= _PLANS.LIMIT:
            plans.pop(next(iter(plans)))
        plan = plans[pkey] = [_lib.lib().pf_conv_gemm_workspace_size(C.byref(d)), None, d, _lib.lib().pf_conv_gemm_kernel_id(C.byref(d))]
    d = plan[2]
    if plan_only:
        return plan[3]
    d.a0, d.a1, d.w, d.bias, d.rowvec, d.residual, d.out = _p(a0), _p(a1), _p(w), _p(bias), _p(rowvec), _p(residual), _p(out)
    d.gn_partial, d.tickets, d.n_tickets = None, None, 0
    nbytes = plan[0]
    ws = torch.empty(nbytes, device=a0.device, dtype=torch.uint8) if nbytes else None   # split-K slabs
    d.workspace, d.workspace_bytes = _p(ws), nbytes
    if nbytes and SPLITK_INKERNEL:
        # a split-K plan, combined inside the launch by the last-arriving workgroup (off by default: it does not pay at these
        # tile sizes, DESIGN.md 11.4 -- and without it no counter ring has to exist before a graph capture)
        d.tickets, d.n_tickets = _ticket_slice(a0.device), _TICKET_SLICE
    gn = None
    if gn_stats and GN_FROM_EPILOGUE and batch == 1:
        if plan[1] is None:
            plan[1] = _lib.lib().pf_conv_gemm_gn_rows(C.byref(d))
        rows = plan[1]
        if rows > 0:
            gn = (torch.empty(M // rows, 2, n_out // 2, device=a0.device, dtype=torch.float32), rows)
            d.gn_partial = _p(gn[0])
    if TRACE is None:
        check(_lib.lib().pf_conv_gemm(C.byref(d), _stream()), "pf_conv_gemm")
    else:
        _traced("k_conv_gemm", 2.0 * M * n_out * (algo_k or ksize * ksize * (c0 + c1)) * batch,
                lambda: check(_lib.lib().pf_conv_gemm(C.byref(d), _stream()), "pf_conv_gemm"),
                "M%d N%d K%d k%d s%d u%d b%d%s" % (M, n_out, ksize * ksize * (c0 + c1), ksize, stride, upsample, batch, " g32" if plan[3] == 2 else " subpixel" if subpixel else ""))
    if gn is not None:
        out._pf_gn = gn                      # (a tensor that carries moments must not be written in place afterwards)
    elif hasattr(out, "_pf_gn"):
        del out._pf_gn                       # a reused out= buffer: moments of what it held before are stale now
    return out


def gemm_workspace_bytes(a0, w, n_out, *, a1=None, n_img=1, h_in=1, w_in=None, ksize=1, stride=1, pad=0,
                         upsample=0, batch=1, wrap_pad=0, crop=0, subpixel=False, **_):
    """Split-K scratch pf_conv_gemm wants for this problem (0: the K range is not split)."""
    d = ConvDesc()
    d.c0, d.c1 = a0.shape[-1], (a1.shape[-1] if a1 is not None else 0)
    d.a1 = _p(a1)
    if w_in is None:
        w_in = a0.numel() // (a0.shape[-1] * max(batch, 1))
    d.n_img, d.ksize, d.batch, d.n_out = n_img, ksize, batch, n_out
    d.wrap_pad, d.crop, d.subpixel = wrap_pad, crop, int(bool(subpixel))
    d.h_out = ((h_in << upsample) + 2 * pad - ksize) // stride + 1
    d.w_out = (((w_in + 2 * wrap_pad) << upsample) + 2 * pad - ksize) // stride + 1 - 2 * crop
    return _lib.lib().pf_conv_gemm_workspace_size(C.byref(d))


# ---- weight-stationary linear (pf_linear_ws): the C = 320 token layers
LWS_16, LWS_F32, LWS_GEGLU, LWS_QKV, LWS_F32_LN, LWS_VT = 0, 1, 2, 3, 4, 5
LINEAR_WS = os.environ.get("PF_LINEAR_WS", "1") != "0"          # A/B: 0 = every linear on the tile kernel (pf_conv_gemm)
LINEAR_WS_K640 = os.environ.get("PF_LINEAR_WS_K640", "1") != "0"   # A/B: 0 = the K = 640 layers (32^2 level: FF1, q | k) stay on the tile kernel
LINEAR_WS_K1280 = os.environ.get("PF_LINEAR_WS_K1280", "1") != "0"  # A/B: 0 = the K = 1280 layers (16^2 level: FF1, q | k, to_q) stay on the tile kernel
LINEAR_VT = os.environ.get("PF_LINEAR_VT", "1") != "0"              # A/B: 0 = V^T projections by the operand-swapped tile GEMM (linear_t)
LINEAR_WS_MIN_ROWS = int(os.environ.get("PF_LINEAR_WS_MIN_ROWS", "8192"))   # fewer 64-token tiles than workgroups: the tile kernel


def linear_ws_ok(rows, N, K, mode, x=None):
    """Does pf_linear_ws serve this problem?  K == 320 (N a multiple of 320, every mode), K == 640 (N a multiple of 256: 16-bit and GEGLU
    outputs; of 128: 16-bit) or K == 1280 (N a multiple of 128: 16-bit and GEGLU), and enough token tiles to stream."""
    if not LINEAR_WS or rows < LINEAR_WS_MIN_ROWS:
        return False
    if K == 640:                  # 256-channel workgroups (16-bit, GEGLU) or 128-channel workgroups (16-bit: to_q, N = 640)
        if not LINEAR_WS_K640 or not ((N % 256 == 0 and mode in (LWS_16, LWS_GEGLU)) or (N % 128 == 0 and mode in (LWS_16, LWS_VT))):
            return False
    elif K == 1280:               # 16^2 level: 128-channel workgroups, 16-token tiles (16-bit, GEGLU)
        if not LINEAR_WS_K1280 or N % 128 or mode not in (LWS_16, LWS_GEGLU, LWS_VT):
            return False
    elif K != 320 or N % 320:
        return False
    elif N == 320 and rows < 4 * LINEAR_WS_MIN_ROWS:     # one channel block: 256 token ranges -- under 2 tiles each the tile kernel wins
        return False                                      # (M = 16384: 0.88 - 0.94 x; q | k | v and FF1 win from 8192 rows, profiles/r4_lws_notes.txt)
    if x is not None and (x.dtype not in (torch.float16, torch.bfloat16) or x.stride(-1) != 1 or x.stride(-2) % 8):
        return False
    return bool(_lib.lib().pf_linear_ws_supported(rows, N, K, mode))


def linear_ws(x, w, mode, bias=None, residual=None, out=None, out_vt=None, rows_per_batch=0, ln=None):
    """pf_linear_ws: x [rows, K] 16-bit, w [N, K] (K = 320; K = 640 for the 16-bit and GEGLU modes).  mode LWS_16 -> [rows, N] 16-bit; LWS_F32 -> fp32 [rows, N] (+ fp32 residual);
    LWS_GEGLU -> [rows, N/2]; LWS_QKV (N = 960) -> ((q | k) [rows, 640], V^T [rows / rows_per_batch, 320, rows_per_batch]);
    LWS_F32_LN (N = 320, ln = (gamma, beta, eps)) -> (fp32 [rows, 320], LayerNorm of it in 16 bit [rows, 320]); LWS_VT (K = 640 / 1280)
    -> the output transposed, [rows / rows_per_batch, N, rows_per_batch]."""
    rows, K = x.shape
    N = w.shape[0]
    d = LinearWsDesc()
    ln_out = None
    if mode == LWS_F32_LN:
        ln_out = torch.empty(rows, N, device=x.device, dtype=x.dtype)
        d.ln_gamma, d.ln_beta, d.ln_eps, d.ln_out, d.ln_ld = _p(ln[0]), _p(ln[1]), float(ln[2]), _p(ln_out), N
    if mode == LWS_QKV:
        nb = rows // rows_per_batch
        if out is None:
            out = torch.empty(rows, 640, device=x.device, dtype=x.dtype)
        if out_vt is None:
            out_vt = torch.empty(nb, 320, rows_per_batch, device=x.device, dtype=x.dtype)
        d.out_vt, d.vt_ld, d.rows_per_batch, d.vt_bs = _p(out_vt), out_vt.stride(1), rows_per_batch, out_vt.stride(0)
    elif mode == LWS_VT:
        nb = rows // rows_per_batch
        if out_vt is None:
            out_vt = torch.empty(nb, N, rows_per_batch, device=x.device, dtype=x.dtype)
        d.out_vt, d.vt_ld, d.rows_per_batch, d.vt_bs = _p(out_vt), out_vt.stride(1), rows_per_batch, out_vt.stride(0)
    elif out is None:
        out = torch.empty(rows, N // 2 if mode == LWS_GEGLU else N, device=x.device,
                          dtype=torch.float32 if mode in (LWS_F32, LWS_F32_LN) else x.dtype)
    d.a, d.a_ld, d.w, d.bias = _p(x), _ld(x), _p(w), _p(bias)
    d.residual, d.res_ld = _p(residual), (_ld(residual) if residual is not None else 0)
    for buf in (out, out_vt):            # caller-supplied buffers: GroupNorm moments they carried described their OLD contents
        if buf is not None:
            _written(buf)
    d.out, d.out_ld = _p(out), (_ld(out) if out is not None else 0)
    d.M, d.N, d.K, d.dtype, d.mode = rows, N, K, dt(x), mode
    if TRACE is None:
        check(_lib.lib().pf_linear_ws(C.byref(d), _stream()), "pf_linear_ws")
    else:
        _traced("k_linear_ws", 2.0 * rows * N * K, lambda: check(_lib.lib().pf_linear_ws(C.byref(d), _stream()), "pf_linear_ws"),
                "M%d N%d K%d mode%d" % (rows, N, K, mode))
    return (out, out_vt) if mode == LWS_QKV else (out, ln_out) if mode == LWS_F32_LN else out_vt if mode == LWS_VT else out


def linear_vt(x, w, n_batch):
    """The V projection of a self-attention written transposed, x [n_batch * nk, K] layer-normed tokens, w [N, K] -> V^T [n_batch, N, nk] (what
    pf_attention reads) -- on the weight-stationary kernel where it serves the shape, or None: the caller then takes the operand-swapped tile GEMM
    (linear_t)."""
    rows, K = x.shape
    nk = rows // n_batch
    if not LINEAR_VT:
        return None
    if nk * n_batch != rows or nk % (16 if K == 1280 else 32 if K == 640 else 64) or not w.is_contiguous() or not linear_ws_ok(rows, w.shape[0], K, LWS_VT, x):
        return None
    return linear_ws(x, w, LWS_VT, rows_per_batch=nk)


def linear_ln(x, w, bias, residual, gamma, beta, eps):
    """out = x w^T + bias + residual (fp32 stream) together with LayerNorm(out) in 16 bit, in ONE launch where pf_linear_ws serves
    the shape (C = 320: the row is complete inside a workgroup) -> (out, ln_out); (out, None) otherwise -- the caller then runs
    ops.layernorm itself."""
    rows, K = x.shape
    if (w.shape[0] == 320 and x.dim() == 2 and w.is_contiguous() and residual is not None and residual.dtype == torch.float32
            and linear_ws_ok(rows, 320, K, LWS_F32_LN, x) and os.environ.get("PF_LINEAR_LN", "1") != "0"):
        return linear_ws(x, w, LWS_F32_LN, bias=bias, residual=residual, ln=(gamma, beta, eps))
    return linear(x, w, bias=bias, residual=residual), None


def linear_qkv(x, wqkv, n_batch):
    """Self-attention projections in one launch: x [n_batch * nk, 320] layer-normed tokens, wqkv [960, 320] = (q | k | v)
    -> ((q | k) [rows, 640], V^T [n_batch, 320, nk]), or None when pf_linear_ws does not serve the shape."""
    rows = x.shape[0]
    nk = rows // n_batch
    if wqkv.shape[0] != 960 or nk % 64 or not linear_ws_ok(rows, 960, x.shape[1], LWS_QKV, x):
        return None
    return linear_ws(x, wqkv, LWS_QKV, rows_per_batch=nk)


def linear(x, w, bias=None, residual=None, out=None, out_dtype=None, geglu=False, split_out=False, gn_stats=False):
    """x [rows, K] 16-bit, w [N, K] 16-bit.  geglu: w / bias rows interleaved (value_j, gate_j),
    returns [rows, N/2] = value * gelu(gate).  gn_stats: see conv_gemm (the moment runs are runs of token rows; the
    consumer's images must be whole runs, which groupnorm_scale_shift's entry point checks).
    The C = 320 layers with enough tokens go to the weight-stationary kernel (pf_linear_ws)."""
    rows, K = x.shape
    if not (split_out or gn_stats) and x.dim() == 2 and w.is_contiguous():
        odt = out.dtype if out is not None else (out_dtype or (residual.dtype if residual is not None else x.dtype))
        mode = None
        if geglu and residual is None and odt == x.dtype:
            mode = LWS_GEGLU
        elif not geglu and odt == torch.float32 and (residual is None or residual.dtype == torch.float32):
            mode = LWS_F32
        elif not geglu and residual is None and odt == x.dtype:
            mode = LWS_16
        if mode is not None and linear_ws_ok(rows, w.shape[0], K, mode, x):
            return linear_ws(x, w, mode, bias=bias, residual=residual, out=out)
    return conv_gemm(x, w, w.shape[0], w_in=rows, bias=bias, residual=residual, out=out, out_dtype=out_dtype,
                     geglu=geglu, split_out=split_out, gn_stats=gn_stats)


def interleave_geglu(w, b=None):
    """[value rows | gate rows] (diffusers / reference GEGLU.proj, transformer.py:8-21) ->
    rows interleaved (value_0, gate_0, value_1, gate_1, ...) for the fused epilogue."""
    inner = w.shape[0] // 2
    wi = torch.stack([w[:inner], w[inner:]], 1).reshape(w.shape).contiguous()
    if b is None:
        return wi
    return wi, torch.stack([b[:inner], b[inner:]], 1).reshape(b.shape).contiguous()


def linear_t(x, w, out=None, ld=None):
    """Transposed projection for attention values: x [B, rows, K], w [N, K] ->
    out [B, N, ld] with out[b, n, r] = sum_k w[n,k] x[b,r,k]   (keys contiguous)."""
    B, rows, K = x.shape
    N = w.shape[0]
    ld = ld or ((rows + 31) // 32) * 32
    if out is None:
        out = torch.empty(B, N, ld, device=x.device, dtype=x.dtype)
    # roles swapped: GEMM rows = weight rows (N of them), GEMM columns = tokens
    conv_gemm(w, x, rows, w_in=N, out=out, out_ld=ld, batch=B,
              a_bstride=0, w_bstride=rows * K, out_bstride=N * ld, c0=K, a0_ld=K)
    return out


def conv_in(x, wgt, bias, cout, dtype, wrap=False, out=None):
    x = x.contiguous()
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, h, w, cout, device=x.device, dtype=dtype)
    check(_lib.lib().pf_conv_in(_p(x), n, cin, h, w, _p(wgt), _p(bias), cout, int(wrap), dt(dtype), _p(out), _stream()),
          "pf_conv_in")
    return out


def conv_out(x, wgt, bias, cout, wrap=False, out=None):
    n, h, w, cin = x.shape
    if out is None:
        out = torch.empty(n, cout, h, w, device=x.device, dtype=torch.float32)
    check(_lib.lib().pf_conv_out(_p(x), dt(x), n, cin, h, w, _p(wgt), _p(bias), cout, int(wrap), _p(out), _stream()),
          "pf_conv_out")
    return out


def conv_out_weight_t(weight):
    """conv_out weight [cout <= 4, cin, 3, 3] -> fp32 [3, 3, cin, 4] (output channels padded with zeros): conv_out_gn's layout."""
    cout, cin = weight.shape[:2]
    wt = torch.zeros(3, 3, cin, 4, device=weight.device, dtype=torch.float32)
    wt[..., :cout] = weight.detach().float().permute(2, 3, 1, 0)
    return wt.contiguous()


def conv_out_gn(x, scale, shift, act, wgt_t, bias, cout, wrap=False, out=None):
    """conv_out(act(x * scale + shift)) in one launch: x fp32 [n, h, w, cin] before the GroupNorm, scale / shift [n, cin]."""
    n, h, w, cin = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and cin % 32 == 0 and cout <= 4
    if out is None:
        out = torch.empty(n, cout, h, w, device=x.device, dtype=torch.float32)
    check(_lib.lib().pf_conv_out_gn(_p(x), n, cin, h, w, _p(scale), _p(shift), int(act), _p(wgt_t), _p(bias), cout, int(wrap),
                                    _p(out), _stream()), "pf_conv_out_gn")
    return out


# ---------------------------------------------------------------------------- attention
def attention(q, k, vt, B, H, D, nq, nk, *, q_ld, k_ld, vt_ld, o_ld=None, q_bs, k_bs, vt_bs, o_bs=None,
              scale=None, bias=None, flags=None, out=None, lse=None):
    """lse: optional fp32 [B, H, nq] output (log2-domain log-sum-exp of the rows) for attention_bwd."""
    o_ld = o_ld or H * D
    o_bs = o_bs if o_bs is not None else nq * o_ld
    if out is None:
        out = torch.empty(B, nq, o_ld, device=q.device, dtype=q.dtype)
    d = AttnDesc()
    d.q, d.k, d.vt, d.out = _p(q), _p(k), _p(vt), _p(out)
    d.dtype, d.B, d.H, d.D, d.nq, d.nk = dt(q), B, H, D, nq, nk
    d.q_ld, d.k_ld, d.vt_ld, d.o_ld = q_ld, k_ld, vt_ld, o_ld
    d.q_bs, d.k_bs, d.vt_bs, d.o_bs = q_bs, k_bs, vt_bs, o_bs
    d.scale = scale if scale is not None else D ** -0.5
    d.bias, d.bias_ld = _p(bias), (_ld(bias) if bias is not None else 0)
    d.flags, d.flags_ld = _p(flags), (_ld(flags) if flags is not None else 0)
    d.lse = _p(lse)
    d.workspace, d.workspace_bytes = None, 0
    if bias is not None and lse is None:
        # few query blocks, many keys (the panorama-query direction of an EPA block): scratch that lets the library split the key range over more
        # workgroups (pf_attn_desc.workspace; 0 bytes = this problem is not split)
        nbytes = _lib.lib().pf_attention_workspace_size(C.byref(d))
        if nbytes:
            ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
            d.workspace, d.workspace_bytes = _p(ws), nbytes
    _traced("k_attention", 4.0 * B * H * nq * nk * D,
            lambda: check(_lib.lib().pf_attention(C.byref(d), _stream()), "pf_attention"),
            "B%d H%d D%d nq%d nk%d bias%d" % (B, H, D, nq, nk, bias is not None))
    return out


# ---------------------------------------------------------------------------- training (EPA block backward)
def transpose_tokens(x, out=None):
    """x [B, T, C] (contiguous, 16-bit or fp32) -> [B, C, T]: the token-contiguous operand layout of the products whose
    reduction runs over tokens (attention backward, weight gradients)."""
    B, T, Cc = x.shape
    if out is None:
        out = torch.empty(B, Cc, T, device=x.device, dtype=x.dtype)
    if x.dtype == torch.float32:
        check(_lib.lib().pf_nhwc_to_nchw(_p(x), dt(x), B, Cc, 1, T, dt(out), _p(out), _stream()), "pf_nhwc_to_nchw")
    else:
        check(_lib.lib().pf_transpose_tokens(_p(x), dt(x), B, T, Cc, _p(out), _stream()), "pf_transpose_tokens")
    return out


def attention_delta(out, dout, B, H, D, nq):
    """rowsum(dout * out) per head: out / dout [B, nq, H*D] contiguous 16-bit -> fp32 [B, H, nq]."""
    delta = torch.empty(B, H, nq, device=out.device, dtype=torch.float32)
    check(_lib.lib().pf_attention_delta(_p(out), _p(dout), dt(out), B, H, D, nq, H * D, nq * H * D, _p(delta), _stream()),
          "pf_attention_delta")
    return delta


def attention_bwd(q, k, v, dout, qt, kt, dot, lse, delta, dq, dk, dv, B, H, D, nq, nk, *, q_ld, k_ld, v_ld, do_ld,
                  dq_ld, dk_ld, dv_ld, q_bs, k_bs, v_bs, do_bs, dq_bs, dk_bs, dv_bs, scale=None, bias=None, flags=None):
    """Backward of attention(): row-major q / k / v / dout (column views allowed: leading dimension + batch stride),
    transposed qt / kt / dot [B, H*D, tokens] (tokens contiguous; row slices of a wider transpose allowed),
    lse / delta fp32 [B, H, nq]; writes dq, dk, dv."""
    assert qt.dim() == 3 and kt.dim() == 3 and dot.dim() == 3 and qt.stride(-1) == kt.stride(-1) == dot.stride(-1) == 1
    d = AttnBwdDesc()
    d.q, d.k, d.v, d.dout = _p(q), _p(k), _p(v), _p(dout)
    d.qt, d.kt, d.dot = _p(qt), _p(kt), _p(dot)
    d.dq, d.dk, d.dv = _p(dq), _p(dk), _p(dv)
    d.dtype, d.B, d.H, d.D, d.nq, d.nk = dt(q), B, H, D, nq, nk
    d.q_ld, d.k_ld, d.v_ld, d.do_ld = q_ld, k_ld, v_ld, do_ld
    d.qt_ld, d.kt_ld, d.dot_ld = qt.stride(-2), kt.stride(-2), dot.stride(-2)      # (row slices of a wider transpose)
    d.dq_ld, d.dk_ld, d.dv_ld = dq_ld, dk_ld, dv_ld
    d.q_bs, d.k_bs, d.v_bs, d.do_bs = q_bs, k_bs, v_bs, do_bs
    d.qt_bs, d.kt_bs, d.dot_bs = qt.stride(0), kt.stride(0), dot.stride(0)
    d.dq_bs, d.dk_bs, d.dv_bs = dq_bs, dk_bs, dv_bs
    d.scale = scale if scale is not None else D ** -0.5
    d.bias, d.bias_ld = _p(bias), (_ld(bias) if bias is not None else 0)
    d.flags, d.flags_ld = _p(flags), (_ld(flags) if flags is not None else 0)
    d.lse, d.delta = _p(lse), _p(delta)
    d.workspace, d.workspace_bytes = None, 0
    nbytes = _lib.lib().pf_attention_bwd_workspace_size(C.byref(d))
    if nbytes:
        ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = _p(ws), nbytes
    _traced("k_attention_bwd", 10.0 * B * H * nq * nk * D,
            lambda: check(_lib.lib().pf_attention_bwd(C.byref(d), _stream()), "pf_attention_bwd"),
            "B%d H%d D%d nq%d nk%d bias%d" % (B, H, D, nq, nk, bias is not None))


def colsum(x, out=None):
    """x [rows, N] (16-bit or fp32, rows may be strided) -> fp32 [N] column sums in a fixed order."""
    rows, N = x.shape
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    nbytes = _lib.lib().pf_colsum_workspace_size(rows, N)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(_lib.lib().pf_colsum(_p(x), dt(x), rows, N, _ld(x), _p(out), _p(ws), nbytes, _stream()), "pf_colsum")
    return out


def layernorm_bwd(x, gamma, dy, eps=1e-5, pe=None, dres=None, dx=None):
    """Backward of layernorm(x + pe): x [rows, C] (16-bit or fp32), dy fp32 [rows, C] (gradient of the normalised,
    affine output), dres optional fp32 gradient added to dx.  Returns (dx fp32, dgamma fp32 [C], dbeta fp32 [C])."""
    rows, Cc = x.shape
    if dx is None:
        dx = torch.empty(rows, Cc, device=x.device, dtype=torch.float32)
    parts = _lib.lib().pf_layernorm_bwd_parts(rows)
    partials = torch.empty(2, parts, Cc, device=x.device, dtype=torch.float32)
    check(_lib.lib().pf_layernorm_bwd(_p(x), _p(pe), 0 if pe is None else pe.shape[0], dt(x), rows, Cc, _p(gamma), eps,
                                      _p(dy), _p(dres), _p(dx), _p(partials), _stream()), "pf_layernorm_bwd")
    return dx, colsum(partials[0]), colsum(partials[1])


def geglu_bwd(u, dg, out=None):
    """u [rows, 2*inner] = [a | gate] (input of geglu), dg [rows, inner] -> du [rows, 2*inner]."""
    rows, two_inner = u.shape
    if out is None:
        out = torch.empty_like(u)
    check(_lib.lib().pf_geglu_bwd(_p(u), _p(dg), dt(u), rows, two_inner // 2, _p(out), _stream()), "pf_geglu_bwd")
    return out


def grad_scale_state(tensors):
    """Device-side power-of-two normalisation of a set of fp32 gradients (no host synchronisation): returns the
    4-float state tensor; state[1] = 2^-e with max|g| * 2^-e in [1, 2), state[2] = 2^e."""
    state = torch.empty(4, device=tensors[0].device, dtype=torch.float32)
    for i, t in enumerate(tensors):
        check(_lib.lib().pf_amax_f32(_p(t), t.numel(), _p(state), int(i == 0), _stream()), "pf_amax_f32")
    check(_lib.lib().pf_pow2_scale(_p(state), _stream()), "pf_pow2_scale")
    return state


def scale_by_state(x, state, index, out_dtype=torch.float32, out=None):
    """y = x * state[index] (x fp32 contiguous); out may alias x when fp32."""
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    else:
        _written(out)
    check(_lib.lib().pf_scale_f32(_p(x), x.numel(), _p(state), index, dt(out), _p(out), _stream()), "pf_scale_f32")
    return out


def groupnorm_bwd(x0, x1, n_img, hw, groups, eps, gamma, scale, shift, act, dy, dres=None):
    """Backward of scale_shift_act(groupnorm_scale_shift(..)): x0 [n, hw, c0] (+ x1 [n, hw, c1]), dy fp32 [n, hw, c0 + c1]
    -> (dx0, dx1) fp32 (dx1 None without a second source).  dres: fp32 gradient of the concat added to the result."""
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    dx0 = torch.empty(n_img, hw, c0, device=x0.device, dtype=torch.float32)
    dx1 = torch.empty(n_img, hw, c1, device=x0.device, dtype=torch.float32) if c1 else None
    nbytes = _lib.lib().pf_groupnorm_bwd_workspace_size(n_img, hw, groups)
    ws = torch.empty(nbytes, device=x0.device, dtype=torch.uint8)
    check(_lib.lib().pf_groupnorm_bwd(_p(x0), c0, _p(x1), c1, dt(x0), n_img, hw, groups, eps, _p(gamma), _p(scale), _p(shift),
                                      int(act), _p(dy), _p(dres), _p(dx0), _p(dx1), _p(ws), nbytes, _stream()), "pf_groupnorm_bwd")
    return dx0, dx1


def groupnorm_param_grads(x0, x1, n_img, hw, scale, shift, unit_scale, unit_shift, act, dy):
    """(dgamma, dbeta) fp32 [c0 + c1] of a trainable GroupNorm (+ SiLU) whose forward was scale_shift_act(x, scale, shift):
    unit_scale / unit_shift = groupnorm_scale_shift of the same input with gamma = 1, beta = 0 (xhat = x * unit_scale + unit_shift)."""
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    Cc = c0 + c1
    out = torch.empty(2 * Cc, device=x0.device, dtype=torch.float32)
    nbytes = _lib.lib().pf_groupnorm_param_grads_workspace_size(n_img, hw, Cc)
    ws = torch.empty(nbytes, device=x0.device, dtype=torch.uint8)
    check(_lib.lib().pf_groupnorm_param_grads(_p(x0), c0, _p(x1), c1, dt(x0), n_img, hw, _p(scale), _p(shift), _p(unit_scale),
                                              _p(unit_shift), int(act), _p(dy), _p(out), _p(ws), nbytes, _stream()),
          "pf_groupnorm_param_grads")
    return out[:Cc], out[Cc:]


def silu_bwd(z, dy):
    """dy * silu'(z): z 16-bit or fp32 (any shape, contiguous), dy fp32 of the same shape -> fp32."""
    assert z.is_contiguous() and dy.is_contiguous() and dy.dtype == torch.float32 and z.shape == dy.shape
    out = torch.empty_like(dy)
    check(_lib.lib().pf_silu_bwd(_p(z), dt(z), _p(dy), z.numel(), _p(out), _stream()), "pf_silu_bwd")
    return out


def im2col3(x, stride=1):
    """x NHWC 16-bit [n, h, w, C] -> [n * ho * wo, 9 * C]: the nine zero-padded taps of a 3x3 / pad 1 convolution (tap-major,
    channels fastest: the column order of the packed conv weights [cout, ky, kx, cin])."""
    n, h, w, Cc = x.shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty(n * ho * wo, 9 * Cc, device=x.device, dtype=x.dtype)
    check(_lib.lib().pf_im2col3(_p(x), dt(x), n, h, w, Cc, stride, _p(out), _stream()), "pf_im2col3")
    return out


def lora_fold(w, up, down, scale, out, out_t=None, d_out=None, u_out=None):
    """out [N, K] (16-bit, rows may be a slice of a packed weight) = w + scale * up @ down from fp32 w [N, K], up [N, r],
    down [r, K] (up = down = None: a plain conversion); optional by-products out_t [K, >= N] (transpose), d_out [r, >= K] (down)
    and u_out [r, >= N] (up^T) in the same 16-bit type, each a (possibly column-offset) view with contiguous rows."""
    N, K = w.shape
    r = 0 if up is None else up.shape[1]
    for t_ in (w, up, down):
        assert t_ is None or (t_.dtype == torch.float32 and t_.is_contiguous() and t_.device == out.device)
    for t_ in (out, out_t, d_out, u_out):
        assert t_ is None or (t_.stride(-1) == 1 and t_.dtype == out.dtype)
    ld = lambda t_: 0 if t_ is None else t_.stride(0)
    check(_lib.lib().pf_lora_fold(_p(w), _p(up), _p(down), N, K, r, float(scale), dt(out), _p(out), ld(out), _p(out_t), ld(out_t),
                                  _p(d_out), ld(d_out), _p(u_out), ld(u_out), _stream()), "pf_lora_fold")
    return out


def weighted_colsum(x, w, dev_scale=None, host_scale=1.0, blocks=None):
    """sum_t w[r, t] * x[t, c]: x [T, C] 16-bit (rows may be strided), w [R, T] fp32 (R a multiple of 4; rows go through the kernel in chunks of 16) -> fp32 [R, C]; with blocks
    = [(row0, rows, col0, cols), ...] (<= 4) a flat fp32 tensor holding only those blocks, each TRANSPOSED ([cols, rows]), one after
    the other.  dev_scale: optional 1-element fp32 device tensor multiplied into the result together with host_scale."""
    T, Cc = x.shape
    R = w.shape[0]
    assert w.dtype == torch.float32 and w.stride(1) == 1 and w.shape[1] >= T and x.stride(1) == 1
    if blocks:
        arr = (C.c_int * (4 * len(blocks)))(*[int(v) for b in blocks for v in b])
        out = torch.empty(sum(b[1] * b[3] for b in blocks), device=x.device, dtype=torch.float32)
    else:
        arr = None
        out = torch.empty(R, Cc, device=x.device, dtype=torch.float32)
    nbytes = _lib.lib().pf_weighted_colsum_workspace_size(T, Cc, R)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(_lib.lib().pf_weighted_colsum(_p(x), dt(x), T, Cc, x.stride(0), _p(w), R, w.stride(0), _p(dev_scale), float(host_scale),
                                        arr, len(blocks) if blocks else 0, _p(out), _p(ws), nbytes, _stream()), "pf_weighted_colsum")
    return out


def zero_insert2(x):
    """x NHWC 16-bit [n, h, w, C] -> [n, 2h, 2w, C] with x at the even positions (stride-2 conv data gradient)."""
    n, h, w, Cc = x.shape
    out = torch.empty(n, 2 * h, 2 * w, Cc, device=x.device, dtype=x.dtype)
    check(_lib.lib().pf_zero_insert2(_p(x), dt(x), n, h, w, Cc, _p(out), _stream()), "pf_zero_insert2")
    return out


def sum2x2(x):
    """x fp32 NHWC [n, 2h, 2w, C] -> [n, h, w, C] block sums (nearest x2 up-sampling backward)."""
    n, h2, w2, Cc = x.shape
    out = torch.empty(n, h2 // 2, w2 // 2, Cc, device=x.device, dtype=torch.float32)
    check(_lib.lib().pf_sum2x2(_p(x), n, h2 // 2, w2 // 2, Cc, _p(out), _stream()), "pf_sum2x2")
    return out


def pad_width_bwd(dy, pad):
    """Gradient of pad_width: dy fp32 [n, h, w + 2 pad, C] -> [n, h, w, C]."""
    n, h, wp, Cc = dy.shape
    out = torch.empty(n, h, wp - 2 * pad, Cc, device=dy.device, dtype=torch.float32)
    check(_lib.lib().pf_pad_width_bwd(_p(dy), n, h, wp - 2 * pad, Cc, pad, _p(out), _stream()), "pf_pad_width_bwd")
    return out


def crop_width_bwd(dy, crop):
    """Gradient of crop_width: dy fp32 [n, h, w - 2 crop, C] -> [n, h, w, C] (zero margins)."""
    n, h, wc, Cc = dy.shape
    out = torch.empty(n, h, wc + 2 * crop, Cc, device=dy.device, dtype=torch.float32)
    check(_lib.lib().pf_crop_width_bwd(_p(dy), n, h, wc + 2 * crop, Cc, crop, _p(out), _stream()), "pf_crop_width_bwd")
    return out
