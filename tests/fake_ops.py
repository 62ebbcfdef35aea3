"""CPU test double for ``panfusion_amd.ops`` (TEST INFRASTRUCTURE, never shipped).

Implements every entry of the ops front end with plain fp32 torch math on CPU tensors, honouring
the same calling conventions (views, leading dimensions, out= buffers).  It lets the CPU suite
exercise the HOST logic of the product -- layer sequencing, skip/pad/crop bookkeeping, weight
packing, sharding and collectives over gloo -- without a GPU.  It is not a fallback: the product
never imports it, and nothing here is measured.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import geometry as G

TRACE = None


def dt(t):
    return {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[t.dtype if isinstance(t, torch.Tensor) else t]


def _cams(fov, theta, phi):
    host = lambda v: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float64).reshape(-1)
    f, t, p = host(fov), host(theta), host(phi)
    n = max(len(f), len(t), len(p))
    return [np.broadcast_to(a, (n,)) for a in (f, t, p)]


def e2p_grid(fov, theta, phi, eh, ew, h, w, device, want_lonlat=False):
    f, t, p = _cams(fov, theta, phi)
    maps = [G.e2p_grid(eh, ew, f[i], t[i], p[i], h, w) for i in range(len(f))]
    mx = torch.tensor(np.stack([m[0] for m in maps]), dtype=torch.float32)
    my = torch.tensor(np.stack([m[1] for m in maps]), dtype=torch.float32)
    if not want_lonlat:
        return mx, my
    ll = [np.stack(G.pers_lonlat(f[i], t[i], p[i], h, w), -1) for i in range(len(f))]
    return mx, my, torch.tensor(np.stack(ll), dtype=torch.float32)


def p2e_grid(fov, theta, phi, ph, pw, h, w, device):
    f, t, p = _cams(fov, theta, phi)
    maps = [G.p2e_grid(ph, pw, f[i], t[i], p[i], h, w) for i in range(len(f))]
    tt = lambda k, d: torch.tensor(np.stack([m[k] for m in maps]), dtype=d)
    return tt(0, torch.float32), tt(1, torch.float32), tt(2, torch.uint8)


def remap(src, map_x, map_y, mode, mask=None):
    from oracle import third_party as tp
    out = tp.remap(src.float(), map_x, map_y, align_corners=True, mode=mode)
    if mask is not None:
        out = out * mask.unsqueeze(1)
    return out.to(src.dtype)


def equi_coords(H, W, device):
    lon, lat = np.linspace(-np.pi, np.pi, W), np.linspace(np.pi / 2, -np.pi / 2, H)
    return torch.tensor(np.stack(np.broadcast_arrays(lon[None, :], lat[:, None]), -1), dtype=torch.float32)


def spherical_pe(coords, freq_bands):
    return G.spherical_pe(coords, freq_bands)


def epa_tables(fov, theta, phi, ph, pw, eh, ew, device):
    f, t, p = _cams(fov, theta, phi)
    cams = {"FoV": torch.tensor(f), "theta": torch.tensor(t), "phi": torch.tensor(p)}
    pm, em = G.get_masks(ph, pw, eh, ew, cams)
    m, E, P = len(f), eh * ew, ph * pw
    bias_e = (pm.reshape(m, E, P).permute(1, 0, 2).reshape(E, m * P) + 1).contiguous()
    bias_p = (em.reshape(m * P, E) + 1).contiguous()

    def flags(b):
        nq, nk = b.shape
        pad = F.pad(b, (0, (-nk) % 32, 0, (-nq) % 32))
        return (pad.reshape(pad.shape[0] // 32, 32, pad.shape[1] // 32, 32).abs().amax((1, 3)) > 0).to(torch.uint8)
    return bias_e, bias_p, flags(bias_e), flags(bias_p)


def _cat(x0, x1):
    return x0 if x1 is None else torch.cat([x0, x1], -1)


def carry(dst, src):
    st = getattr(src, "_pf_gn", None)
    if st is not None:
        dst._pf_gn = st
    return dst


GN_FROM_PARTIALS = [0, 0]       # [statistics taken from GEMM-epilogue moments, statistics by a pass over the tensor] (test probes)


def groupnorm_scale_shift(x0, x1, n_img, hw, groups, eps, gamma, beta, ws=None, wrap=None):
    if wrap is not None and wrap[1] > 0:          # statistics of the circularly padded tensor
        w, p = wrap
        x = _cat(x0, x1).float().reshape(n_img, hw // w, w, -1)
        x = torch.cat([x[:, :, -p:], x, x[:, :, :p]], 2)
        return groupnorm_scale_shift(x.reshape(n_img, -1, x.shape[-1]), None, n_img, (hw // w) * (w + 2 * p), groups, eps, gamma, beta)
    st0 = getattr(x0, "_pf_gn", None)
    st1 = getattr(x1, "_pf_gn", None) if x1 is not None else None
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    usable = st0 is not None and (x1 is None or st1 is not None) and ((c0 + c1) // groups) % 2 == 0 and c0 % 2 == 0
    if usable and (hw % st0[1] or (st1 is not None and hw % st1[1])):
        usable = False
    if usable:
        # the product's path: per-column-PAIR moments over runs of `rows` output rows, as pf_conv_gemm's epilogue leaves them
        GN_FROM_PARTIALS[0] += 1
        s, q = [], []
        for st in (st0, st1):
            if st is None:
                continue
            part, rows = st
            assert hw % rows == 0 and part.shape[0] == n_img * hw // rows
            pp = part.double().reshape(n_img, hw // rows, 2, -1).sum(1)            # [n, 2, c/2]
            s.append(pp[:, 0])
            q.append(pp[:, 1])
        s, q = torch.cat(s, -1), torch.cat(q, -1)                                  # [n, C/2]
        C = 2 * s.shape[-1]
        cpg2 = C // groups // 2
        sg, qg = s.reshape(n_img, groups, cpg2).sum(-1), q.reshape(n_img, groups, cpg2).sum(-1)
        cnt = float(hw * (C // groups))
        mean = sg / cnt
        var = (qg / cnt - mean * mean).clamp_min(0)
        rstd = (var + eps).rsqrt()
        scale = (rstd.repeat_interleave(C // groups, 1) * gamma.double()).float()
        shift = (beta.double() - mean.repeat_interleave(C // groups, 1) * scale.double()).float()
        return scale, shift
    GN_FROM_PARTIALS[1] += 1
    x = _cat(x0, x1).float().reshape(n_img, hw, -1)
    C = x.shape[-1]
    xg = x.reshape(n_img, hw, groups, C // groups)
    mean = xg.mean((1, 3))
    var = xg.var((1, 3), unbiased=False)
    rstd = (var + eps).rsqrt()
    scale = rstd.repeat_interleave(C // groups, 1) * gamma
    shift = beta - mean.repeat_interleave(C // groups, 1) * scale
    return scale, shift


def _pair(v, dtype):
    """fp32 [.., C] -> the split-precision pair [.., 2C] of the kernels: per block of 32 channels [hi(32) | lo(32)]."""
    hi = v.to(dtype)
    lo = (v - hi.float()).to(dtype)
    C = v.shape[-1]
    return torch.stack([hi.reshape(*v.shape[:-1], C // 32, 32), lo.reshape(*v.shape[:-1], C // 32, 32)], -2).reshape(*v.shape[:-1], 2 * C)


def _unpair(p):
    """pair [.., 2C] -> (hi, lo) [.., C] fp32."""
    C2 = p.shape[-1]
    q = p.float().reshape(*p.shape[:-1], C2 // 64, 2, 32)
    return q[..., 0, :].reshape(*p.shape[:-1], C2 // 2), q[..., 1, :].reshape(*p.shape[:-1], C2 // 2)


def scale_shift_act(x0, x1, n_img, hw, scale, shift, act, out=None, out_dtype=None, split=False, raw_pair=False):
    x = _cat(x0, x1).float().reshape(n_img, hw, -1)
    y = x if scale is None else x * scale[:, None] + shift[:, None]
    y = F.silu(y) if act else y
    out_dtype = out_dtype or x0.dtype
    if raw_pair:
        return y.to(out_dtype), _pair(x, out_dtype)
    if split:
        return _pair(y, out_dtype)
    return y.to(out_dtype)


def layernorm(x, gamma, beta, eps=1e-5, pe=None, out=None, out_dtype=None):
    v = x.float()
    if pe is not None:
        v = v + pe.repeat(x.shape[0] // pe.shape[0], 1)
    y = F.layer_norm(v, (x.shape[-1],), gamma, beta, eps).to(out.dtype if out is not None else (out_dtype or x.dtype))
    if out is not None:
        out.copy_(y)
        return out
    return y


def geglu(x, out=None):
    a, g = x.float().chunk(2, -1)
    return (a * F.gelu(g)).to(x.dtype)


def timestep_features(t, dim, dtype):
    half = dim // 2
    e = (-9.210340371976184 * torch.arange(half, dtype=torch.float32)) / half
    arg = t.float()[:, None] * torch.exp(e)[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], -1).to(dtype)


def silu(x, out=None):
    y = F.silu(x.float()).to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def add(a, b, out=None):
    y = (a.float() + b.float()).to(a.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def pad_width(x, pad, out=None):
    return torch.cat([x[:, :, -pad:], x, x[:, :, :pad]], 2).contiguous()


def crop_width(x, crop, out=None):
    return x[:, :, crop:-crop].contiguous()


def pad_width_rows(x, pad):
    return torch.cat([x[..., -pad:], x, x[..., :pad]], -1).contiguous()


def roll_width(x, shift, out=None):
    y = torch.roll(x, shift, -1)
    if out is not None:
        out.copy_(y)
        return out
    return y


def nchw_to_nhwc(x, dtype, out=None):
    return x.permute(0, 2, 3, 1).to(dtype).contiguous()


def nhwc_to_nchw(x, dtype, out=None):
    return x.permute(0, 3, 1, 2).to(dtype).contiguous()


def embed_tokens(ids, tok, pos, Lp, out_dtype):
    B, L = ids.shape
    out = torch.zeros(B, Lp, tok.shape[1])
    out[:, :L] = tok[ids] + pos[:L]
    return out.to(out_dtype)


def softmax_rows(scores, scale, out_dtype, out=None):
    return torch.softmax(scores.float() * scale, -1).to(out_dtype)


def axpby(x, y, a, b, out=None):
    r = a * x.float() + b * y.float()
    if out is not None:
        out.copy_(r)
        return out
    return r


def vae_sample(moments, eps, scale):
    L = moments.shape[-1] // 2
    m = moments.permute(0, 3, 1, 2)
    return (m[:, :L] + torch.exp(0.5 * m[:, L:].clamp(-30, 20)) * eps) * scale


def tensor_to_image(x):
    return ((x.float() / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def cfg_ddim_step(x, eps_uncond, eps_cond, guidance, coef, roll=0, out=None):
    sa, sb, sap, sbp = coef
    eps = eps_uncond + guidance * (eps_cond - eps_uncond)
    y = torch.roll(sap * ((x - sb * eps) / sa) + sbp * eps, roll, -1)
    if out is not None:
        out.copy_(y)
        return out
    return y


def cfg_ddim_step_pair(x, eps_uncond, eps_cond, guidance, coef, roll=0, out=None, out2=None, tstep=None, t_next=0):
    y = cfg_ddim_step(x, eps_uncond, eps_cond, guidance, coef, roll)
    out = y if out is None else out.copy_(y)
    if out2 is not None:
        out2.copy_(y)
    if tstep is not None:
        tstep.fill_(int(t_next))
    return out


def conv_gemm(a0, w, n_out, *, a1=None, n_img=1, h_in=1, w_in=None, ksize=1, stride=1, pad=0, upsample=0,
              bias=None, rowvec=None, residual=None, out=None, out_dtype=None, batch=1, geglu=False, c0=None, c1=None,
              a0_ld=None, a1_ld=None, algo_k=None, a_bstride=0, w_bstride=0, out_bstride=0, split_out=False, pad_hi=0,
              gn_stats=False, wrap_pad=0, crop=0, split3=False, subpixel=False, **kw):
    if split3:
        # split-precision walk (pf_conv_desc.split3): a0 is a pair tensor, w holds per tap and 32-channel block [W_hi | W_lo];
        # W_hi A_hi + W_hi A_lo + W_lo A_hi  (W_lo A_lo is not formed)
        assert a1 is None and batch == 1
        C2 = c0 or a0.shape[-1]
        a_hi, a_lo = _unpair(a0.reshape(-1, a0_ld or a0.shape[-1])[:, :C2])
        wrows = 4 * n_out if subpixel else n_out                   # (sub-pixel form: [4 phases x n_out] rows of 4 taps)
        w_hi, w_lo = _unpair(w.reshape(wrows, 4 if subpixel else ksize * ksize, C2))
        common = dict(n_img=n_img, h_in=h_in, w_in=w_in, ksize=ksize, stride=stride, pad=pad, upsample=upsample, pad_hi=pad_hi,
                      wrap_pad=wrap_pad, crop=crop, out_dtype=torch.float32, subpixel=subpixel)
        part = conv_gemm(a_hi, w_lo.reshape(wrows, -1), n_out, **common)
        if w_in is None:
            w_in = a_hi.shape[0]
        if subpixel:                                               # (no residual operand in that form: add the W_lo A_hi part afterwards)
            y = conv_gemm(a_hi + a_lo, w_hi.reshape(wrows, -1), n_out, bias=bias, **common) + part
            y = y.to(out_dtype or a0.dtype)
            if out is not None:
                out.copy_(y.reshape(out.shape))
                return out
            return y
        return conv_gemm(a_hi + a_lo, w_hi.reshape(n_out, -1), n_out, bias=bias, rowvec=rowvec,
                         residual=part if residual is None else part + residual.float().reshape(part.shape[0], -1)[:, :n_out],
                         out=out, out_dtype=out_dtype or (residual.dtype if residual is not None else a0.dtype), geglu=geglu,
                         split_out=split_out, gn_stats=gn_stats, **{k: v for k, v in common.items() if k != "out_dtype"})
    if subpixel:
        # pf_conv_desc.subpixel: nearest x2 + 3x3 as four 2x2 phase convolutions on the low-resolution grid; w = [4][n_out][2][2][C].
        # Output pixel (2 y + a, 2 x + b) reads rows y - 1 + a, y + a and columns x - 1 + b, x + b (zero outside the image).
        assert ksize == 3 and upsample == 1 and stride == 1 and pad == 1 and residual is None and rowvec is None and not geglu
        x = a0.float().reshape(-1, a0_ld or a0.shape[-1])[:, :(c0 or a0.shape[-1])]
        if a1 is not None:
            x = torch.cat([x, a1.float().reshape(-1, a1_ld or a1.shape[-1])[:, :(c1 or a1.shape[-1])]], -1)
        C = x.shape[-1]
        x = x.reshape(n_img, h_in, w_in, C).permute(0, 3, 1, 2)
        if wrap_pad:
            x = torch.cat([x[..., -wrap_pad:], x, x[..., :wrap_pad]], -1)
        w4 = w.float().reshape(4, n_out, 2, 2, C)
        y = torch.zeros(n_img, n_out, 2 * x.shape[2], 2 * x.shape[3])
        for a in range(2):
            for b in range(2):
                xp = F.pad(x, (1 - b, b, 1 - a, a))
                y[:, :, a::2, b::2] = F.conv2d(xp, w4[2 * a + b].permute(0, 3, 1, 2))
        if bias is not None:
            y = y + bias.float()[None, :, None, None]
        if crop:
            y = y[..., crop:-crop]
        ho, wo = y.shape[2:]
        y = y.permute(0, 2, 3, 1).reshape(n_img * ho * wo, n_out)
        y = y.to(out_dtype or a0.dtype)
        if out is not None:
            out.copy_(y.reshape(out.shape))
            return out
        return y
    if batch > 1:          # independent problems (attention scores / P.V of the VAE): plain linears only
        assert ksize == 1 and a1 is None and bias is None and residual is None and rowvec is None and not geglu
        K = c0 or a0.shape[-1]
        A = a0.float().reshape(-1)[:a_bstride * (batch - 1) + w_in * K] if a_bstride else None
        outs = []
        for b in range(batch):
            Ab = a0.float().reshape(-1)[b * a_bstride:b * a_bstride + w_in * K].reshape(w_in, K) if a_bstride else a0.float().reshape(w_in, K)
            Wb = w.float().reshape(-1)[b * w_bstride:b * w_bstride + n_out * K].reshape(n_out, K)
            outs.append(Ab @ Wb.t())
        y = torch.stack(outs).to(out_dtype or a0.dtype)
        if out is not None:
            out.copy_(y.reshape(out.shape))
            return out
        return y
    # [pixel][ld] rows of which the first c channels are taken (the split-precision GEMM passes the [hi | lo]
    # pair as source 0 and its hi half again as source 1: same storage, c1 = ld / 2)
    x = a0.float().reshape(-1, a0_ld or a0.shape[-1])[:, :(c0 or a0.shape[-1])]
    if a1 is not None:
        x = torch.cat([x, a1.float().reshape(-1, a1_ld or a1.shape[-1])[:, :(c1 or a1.shape[-1])]], -1)
    C = x.shape[-1]
    if w_in is None:
        w_in = x.shape[0]
    x = x.reshape(n_img, h_in, w_in, C).permute(0, 3, 1, 2)
    if wrap_pad:                                  # virtual pad_pano of the input width (pre-upsampling columns)
        x = torch.cat([x[..., -wrap_pad:], x, x[..., :wrap_pad]], -1)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    wt = w.float().reshape(n_out, ksize, ksize, C).permute(0, 3, 1, 2)
    if pad_hi:
        x = F.pad(x, (0, pad_hi, 0, pad_hi))
    y = F.conv2d(x, wt, None if bias is None else bias.float(), stride=stride, padding=pad)
    if crop:                                      # unpad_pano of the output
        y = y[..., crop:-crop]
    ho, wo = y.shape[2:]
    y = y.permute(0, 2, 3, 1).reshape(n_img * ho * wo, n_out)
    if rowvec is not None:
        y = y + rowvec[:, :n_out].float().repeat_interleave(ho * wo, 0)
    if residual is not None:
        y = y + residual.float().reshape(-1, residual.shape[-1])[:, :n_out]
    gn = None
    if gn_stats and not geglu and not split_out and n_out % 2 == 0:
        rows = next((r for r in (64, 32, 16) if (ho * wo) % r == 0), 0)           # (the kernels: 64 / 32 fragment rows per wavefront)
        if rows:
            yr = y.double().reshape(-1, rows, n_out // 2, 2)
            gn = (torch.stack([yr.sum((1, 3)), (yr * yr).sum((1, 3))], 1).float(), rows)      # [M / rows, 2, N / 2]
    if geglu:                                     # rows interleaved (value_j, gate_j)
        y = y[:, 0::2] * F.gelu(y[:, 1::2])
    if split_out:
        y = _pair(y, a0.dtype)
    else:
        y = y.to(out_dtype or (residual.dtype if residual is not None else a0.dtype))
    if out is not None:
        if out.shape[-1] != y.shape[-1] and out.numel() != y.numel():   # wider row stride: pad columns untouched
            out.view(-1, out.shape[-1])[:, :y.shape[-1]] = y
        else:
            out.copy_(y.reshape(out.shape))
        y = out
    if gn is not None:
        y._pf_gn = gn
    return y


def linear(x, w, bias=None, residual=None, out=None, out_dtype=None, geglu=False, split_out=False, gn_stats=False):
    return conv_gemm(x, w, w.shape[0], w_in=x.shape[0], bias=bias, residual=residual, out=out, out_dtype=out_dtype,
                     geglu=geglu, split_out=split_out, gn_stats=gn_stats)


def linear_ln(x, w, bias, residual, gamma, beta, eps):
    """Test double of ops.linear_ln: the fused form whenever the projection is square (any width: host logic runs on CPU)."""
    out = linear(x, w, bias=bias, residual=residual)
    if w.shape[0] != w.shape[1] or residual is None:
        return out, None
    return out, torch.nn.functional.layer_norm(out.float(), (out.shape[-1],), gamma, beta, eps).to(x.dtype)


def linear_vt(x, w, n_batch):
    return None          # (the test double takes linear_t)


def linear_qkv(x, wqkv, n_batch):
    """Test double of ops.linear_qkv (one launch for q | k | v with V transposed): served whenever the batches are whole
    64-token tiles, at any width -- the host logic that consumes the fused result runs on CPU at the tiny widths too."""
    rows, K = x.shape
    nk = rows // n_batch
    if wqkv.shape[0] != 3 * K or nk % 64:
        return None
    y = (x.float() @ wqkv.float().T).to(x.dtype)
    return y[:, :2 * K].contiguous(), y[:, 2 * K:].reshape(n_batch, nk, K).transpose(1, 2).contiguous()


def interleave_geglu(w, b=None):
    inner = w.shape[0] // 2
    wi = torch.stack([w[:inner], w[inner:]], 1).reshape(w.shape).contiguous()
    if b is None:
        return wi
    return wi, torch.stack([b[:inner], b[inner:]], 1).reshape(b.shape).contiguous()


def linear_t(x, w, out=None, ld=None):
    B, rows, K = x.shape
    ld = ld or ((rows + 31) // 32) * 32
    y = torch.full((B, w.shape[0], ld), float("nan"), dtype=x.dtype)     # padding must never be consumed
    y[:, :, :rows] = torch.einsum("nk,brk->bnr", w.float(), x.float()).to(x.dtype)
    return y


def conv_in(x, wgt, bias, cout, dtype, wrap=False, out=None):
    w = wgt.permute(3, 2, 0, 1)
    if wrap:
        y = F.conv2d(G.pad_pano(x.float(), 1), w, bias, padding=1)[..., 1:-1]
    else:
        y = F.conv2d(x.float(), w, bias, padding=1)
    y = y.permute(0, 2, 3, 1).to(dtype).contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


def conv_out(x, wgt, bias, cout, wrap=False, out=None):
    xi = x.float().permute(0, 3, 1, 2)
    w = wgt.permute(0, 3, 1, 2)
    if wrap:
        return F.conv2d(G.pad_pano(xi, 1), w, bias, padding=1)[..., 1:-1].contiguous()
    return F.conv2d(xi, w, bias, padding=1)


def conv_out_weight_t(weight):
    cout, cin = weight.shape[:2]
    wt = torch.zeros(3, 3, cin, 4, dtype=torch.float32)
    wt[..., :cout] = weight.detach().float().permute(2, 3, 1, 0)
    return wt


def conv_out_gn(x, scale, shift, act, wgt_t, bias, cout, wrap=False, out=None):
    n, h, w, cin = x.shape
    y = x.float() * scale.view(n, 1, 1, cin) + shift.view(n, 1, 1, cin)
    if act:
        y = F.silu(y)
    return conv_out(y, wgt_t[..., :cout].permute(3, 0, 1, 2), bias, cout, wrap=wrap, out=out)


def attention(q, k, vt, B, H, D, nq, nk, *, q_ld, k_ld, vt_ld, o_ld=None, q_bs, k_bs, vt_bs, o_bs=None,
              scale=None, bias=None, flags=None, out=None, lse=None):
    C = H * D
    qh = q.float().reshape(-1, q.shape[-1])[:B * nq, :C].reshape(B, nq, H, D).transpose(1, 2)
    kh = k.float().reshape(-1, k.shape[-1])[:B * nk, :C].reshape(B, nk, H, D).transpose(1, 2)
    v = vt.float().reshape(B, C, -1)[:, :, :nk].reshape(B, H, D, nk).transpose(2, 3)
    s = qh @ kh.transpose(-1, -2) * (scale if scale is not None else D ** -0.5)
    if bias is not None:
        tiles = flags.bool().repeat_interleave(32, 0).repeat_interleave(32, 1)[:nq, :nk]
        assert not bool((bias[:nq, :nk].ne(0) & ~tiles).any()), "non-zero bias outside flagged tiles"
        s = s + bias[:nq, :nk]
    if lse is not None:                               # log2-domain log-sum-exp of the rows
        lse.copy_((torch.logsumexp(s, -1) * 1.4426950408889634).reshape(lse.shape))
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, nq, C).to(q.dtype)
    if out is not None:
        out.copy_(o.reshape(out.shape))
        return out
    return o


# ---------------------------------------------------------------------------- training (EPA block backward)
def transpose_tokens(x, out=None):
    return x.transpose(1, 2).contiguous()


def attention_delta(out, dout, B, H, D, nq):
    return (out.float() * dout.float()).reshape(B, nq, H, D).sum(-1).transpose(1, 2).contiguous()


def attention_bwd(q, k, v, dout, qt, kt, dot, lse, delta, dq, dk, dv, B, H, D, nq, nk, *, scale=None, bias=None, flags=None,
                  **strides):
    heads = lambda t, n: t.float().reshape(B, n, H, D).transpose(1, 2)
    qh, kh, vh, doh = heads(q, nq), heads(k, nk), heads(v, nk), heads(dout, nq)
    # the transposed operands must be the transposes of the row-major ones (the kernel reads both)
    for t, r, n in ((qt, q, nq), (kt, k, nk), (dot, dout, nq)):
        assert torch.equal(t.reshape(B, H * D, n).transpose(1, 2), r.reshape(B, n, H * D))
    scale = scale if scale is not None else D ** -0.5
    s = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias[:nq, :nk]
    p = torch.exp2(s * 1.4426950408889634 - lse.reshape(B, H, nq, 1))
    dvh = p.transpose(-1, -2) @ doh
    ds = p * (doh @ vh.transpose(-1, -2) - delta.reshape(B, H, nq, 1))
    back = lambda t, n: t.transpose(1, 2).reshape(B, n, H * D)
    dq.copy_(back(ds @ kh * scale, nq).to(dq.dtype))
    dk.copy_(back(ds.transpose(-1, -2) @ qh * scale, nk).to(dk.dtype))
    dv.copy_(back(dvh, nk).to(dv.dtype))


def colsum(x, out=None):
    return x.float().sum(0)


def layernorm_bwd(x, gamma, dy, eps=1e-5, pe=None, dres=None, dx=None):
    v = x.float()
    if pe is not None:
        v = v + pe.repeat(x.shape[0] // pe.shape[0], 1)
    mean = v.mean(-1, keepdim=True)
    rstd = (v.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
    xh = (v - mean) * rstd
    g = dy * gamma
    r = rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if dres is not None:
        r = r + dres
    if dx is not None:
        dx.copy_(r)
        r = dx
    return r, (dy * xh).sum(0), dy.sum(0)


def geglu_bwd(u, dg, out=None):
    a, g = u.float().chunk(2, -1)
    cdf = 0.5 * (1 + torch.erf(g * 2 ** -0.5))
    pdf = torch.exp(-0.5 * g * g) * 0.3989422804014327
    d = dg.float()
    return torch.cat([d * g * cdf, d * a * (cdf + g * pdf)], -1).to(u.dtype)


def grad_scale_state(tensors):
    amax = max(float(t.abs().max()) for t in tensors)
    e = int(np.floor(np.log2(amax))) if np.isfinite(amax) and amax > 0 else 0
    return torch.tensor([amax, 2.0 ** -e, 2.0 ** e, 0.0], dtype=torch.float32)


def scale_by_state(x, state, index, out_dtype=torch.float32, out=None):
    y = (x * state[index]).to(out.dtype if out is not None else out_dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def groupnorm_bwd(x0, x1, n_img, hw, groups, eps, gamma, scale, shift, act, dy, dres=None):
    x = _cat(x0.float().reshape(n_img, hw, -1), None if x1 is None else x1.float().reshape(n_img, hw, -1)).clone()
    C = x.shape[-1]
    mu = x.reshape(n_img, hw, groups, C // groups).mean((1, 3), keepdim=True).expand(-1, 1, -1, C // groups).reshape(n_img, C)
    beta = (shift + mu * scale)[0]                   # shift = beta - mean * scale (pf_groupnorm_stats)
    x.requires_grad_(True)
    with torch.enable_grad():
        z = F.group_norm(x.transpose(1, 2), groups, gamma, beta, eps).transpose(1, 2)
        (F.silu(z) if act else z).backward(dy.reshape(z.shape))
    g = x.grad
    if dres is not None:
        g = g + dres.reshape(g.shape)
    c0 = x0.shape[-1]
    return g[..., :c0].contiguous(), (g[..., c0:].contiguous() if x1 is not None else None)


def groupnorm_param_grads(x0, x1, n_img, hw, scale, shift, unit_scale, unit_shift, act, dy):
    x = _cat(x0, x1).float().reshape(n_img, hw, -1)
    z = x * scale[:, None, :] + shift[:, None, :]
    dz = dy.reshape(n_img, hw, -1).float()
    if act:
        sg = torch.sigmoid(z)
        dz = dz * sg * (1 + z * (1 - sg))
    xhat = x * unit_scale[:, None, :] + unit_shift[:, None, :]
    return (dz * xhat).sum((0, 1)), dz.sum((0, 1))


def silu_bwd(z, dy):
    zf = z.float()
    sg = torch.sigmoid(zf)
    return dy * sg * (1 + zf * (1 - sg))


def im2col3(x, stride=1):
    n, h, w, Cc = x.shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))
    taps = [xp[:, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride] for ky in range(3) for kx in range(3)]
    return torch.stack(taps, 3).reshape(n * ho * wo, 9 * Cc)


def lora_fold(w, up, down, scale, out, out_t=None, d_out=None, u_out=None):
    f = w if up is None else w + scale * (up @ down)
    N, K = w.shape
    out.copy_(f.to(out.dtype))
    if out_t is not None:
        out_t[:K, :N].copy_(out.t())
    if d_out is not None:
        d_out[:, :K].copy_(down.to(out.dtype))
    if u_out is not None:
        u_out[:, :N].copy_(up.t().to(out.dtype))
    return out


def weighted_colsum(x, w, dev_scale=None, host_scale=1.0, blocks=None):
    T = x.shape[0]
    full = (w[:, :T].float() @ x.float()) * host_scale * (float(dev_scale[0]) if dev_scale is not None else 1.0)
    if not blocks:
        return full
    return torch.cat([full[r0:r0 + nr, c0:c0 + nc].t().reshape(-1) for r0, nr, c0, nc in blocks])


def zero_insert2(x):
    n, h, w, C = x.shape
    y = torch.zeros(n, 2 * h, 2 * w, C, dtype=x.dtype)
    y[:, ::2, ::2] = x
    return y


def sum2x2(x):
    n, h2, w2, C = x.shape
    return x.reshape(n, h2 // 2, 2, w2 // 2, 2, C).sum((2, 4))


def pad_width_bwd(dy, pad):
    w = dy.shape[2] - 2 * pad
    dx = dy[:, :, pad:pad + w].clone()
    if pad:
        dx[:, :, w - pad:] += dy[:, :, :pad]
        dx[:, :, :pad] += dy[:, :, w + pad:]
    return dx


def crop_width_bwd(dy, crop):
    return F.pad(dy, (0, 0, crop, crop))
