"""Host-side driver of the HIP denoiser: weight packing and layer sequencing.

Takes module trees with the diffusers SD-2 UNet attribute layout (what
``MultiViewBaseModel`` receives in the reference, models/pano/MVGenModel.py:9-36)
and the EPA blocks, repacks their weights ONCE into the kernel layouts
(16-bit, K-contiguous, LoRA folded, q|k projections fused, all time-embedding
projections of a UNet concatenated), and runs the forward as a sequence of
C-ABI calls on the current stream.  Activations stay NHWC / token-major
between kernels; nothing here computes on the host.

Two storage schemes (``precision``):
  * ``"fast"``  -- every activation 16-bit (bf16 / fp16), fp32 accumulation: the round-1 path.
  * ``"mixed"`` -- the scheme that meets north_star's 1e-3 rel-L2 with fp16 operands (error budget:
    tools/precision_study.py, profiles/archive/r2_precision_budget.txt).  MFMA operands stay 16-bit; the residual
    streams (block outputs, the token stream inside a transformer block, the shortcut) are fp32; and the few
    GEMMs that map the stream linearly onto itself -- resnet shortcut 1x1, proj_in, proj_out, the
    downsampling conv -- run as SPLIT-PRECISION GEMMs: A = A_hi + A_lo, W = W_hi + W_lo (16-bit each),
    A_hi W_hi + A_lo W_hi + A_hi W_lo inside one K step of the same kernels (operand and weights interleave
    hi / lo per 32 channels, a 64-element K block holds both of the same channels): ~2^-22 relative, 8 % of
    the step's FLOPs run three MFMAs per operand pair.
"""
import os
from types import SimpleNamespace as NS

import torch

from . import ops


def default_precision(dtype):
    """PF_PRECISION=mixed|fast overrides; otherwise fp16 runs the mixed scheme (the one that passes the
    1e-3 parity bar), bf16 / the fp32 CPU test double the plain 16-bit one."""
    env = os.environ.get("PF_PRECISION")
    if env in ("mixed", "fast"):
        return env
    return "mixed" if dtype == torch.float16 else "fast"


RAW_PAIR_FUSED = os.environ.get("PF_RAW_PAIR", "1") != "0"      # A/B: 0 = the shortcut operand by its own split pass
VIRTUAL_PAD = os.environ.get("PF_VIRTUAL_PAD", "1") != "0"      # A/B: 0 = materialised pad_pano / unpad_pano copies around the panorama convs
EXACT_UP0 = os.environ.get("PF_EXACT_UP0", "1") != "0"          # A/B: 0 = the level-0 upsampling convolution single-pass 16 bit like the other two (mixed scheme)
SUBPIXEL_UP = os.environ.get("PF_SUBPIXEL_UP", "1") != "0"      # A/B: 0 = the upsampling convolutions as nearest x2 + 3x3 (9 taps) instead of four 2x2 phase convolutions (4 taps)
FUSED_HEAD = os.environ.get("PF_FUSED_HEAD", "1") != "0"        # A/B: 0 = GroupNorm-apply + SiLU pass, then conv_out (two launches)


def stream_dtype(dtype, precision):
    return torch.float32 if precision == "mixed" else dtype


# ---------------------------------------------------------------------------- packing helpers
def _f32(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _w16(t, dev, dtype):
    return t.detach().to(device=dev, dtype=torch.float32).to(dtype).contiguous()


def _processor_lora(attn, name):
    """attn.processor.to_{q,k,v,out}_lora of an un-migrated diffusers LoRAAttnProcessor, or None."""
    proc = getattr(attn, "processor", None)
    lora = getattr(proc, name, None) if proc is not None else None
    if lora is not None and not (hasattr(lora, "down") and hasattr(lora, "up")):
        raise TypeError("%s.%s has no down / up matrices" % (type(proc).__name__, name))
    return lora


def _conv3_weight(conv, dev, dtype):
    # torch [Cout, Cin, ky, kx] -> [Cout, ky, kx, Cin] (K ordered tap-major, channel-minor)
    return _w16(conv.weight.detach().float().permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1), dev, dtype)


def _subpixel_weight(conv, dev, dtype):
    """Weights of an upsampling convolution (nearest x2, then 3x3) for the four-phase form of pf_conv_desc.subpixel: output pixel
    (2 y + a, 2 x + b) reads low-resolution rows y - 1 + a, y + a and columns x - 1 + b, x + b only, so per phase the nine taps collapse
    to 2 x 2 with summed weights (rows: a = 0 -> {W[0]}, {W[1] + W[2]}; a = 1 -> {W[0] + W[1]}, {W[2]}; columns alike).  Summed in
    fp32, rounded once.  torch [Cout, Cin, 3, 3] -> [4 phases, Cout, 2, 2, Cin] -> [4 * Cout, 4 * Cin]."""
    w = conv.weight.detach().float()
    groups = (((0,), (1, 2)), ((0, 1), (2,)))
    phases = []
    for a in range(2):
        for b in range(2):
            taps = [[sum(w[:, :, ky, kx] for ky in groups[a][r] for kx in groups[b][c]) for c in range(2)] for r in range(2)]
            phases.append(torch.stack([torch.stack(row, 1) for row in taps], 1))        # [Cout, 2, 2, Cin]
    return _w16(torch.stack(phases).reshape(4 * w.shape[0], -1), dev, dtype)


def _split_weight(w, taps, dev, dtype):
    """Split-precision packing of an fp32 weight [N, taps * C] (tap-major): per tap and per block of 32 channels
    [W_hi(32) | W_lo(32)], the partner of the pair operand of exact_gemm (same interleave)."""
    w = w.detach().to(device=dev, dtype=torch.float32).reshape(w.shape[0], taps, -1, 32)
    hi = w.to(dtype)
    lo = (w - hi.float()).to(dtype)
    return torch.stack([hi, lo], 3).reshape(w.shape[0], -1).contiguous()


def split_operand(x0, x1=None, scale=None, shift=None, act=0, dtype=None):
    """[.., C] stream tensor(s) (channel concat of x0 | x1, optionally GroupNorm-applied) -> [rows, 2C]
    16-bit pair [hi | lo]."""
    c = x0.shape[-1]
    n_img = scale.shape[0] if scale is not None else 1
    hw = x0.numel() // (c * n_img)
    return ops.scale_shift_act(x0, x1, n_img, hw, scale, shift, act, out_dtype=dtype, split=True)


def exact_gemm(a_split, w3, n_out, **kw):
    """Split-precision GEMM / conv: a_split [.., 2C] and w3 (per tap) hold, per block of 32 channels, [hi(32) | lo(32)]; every
    64-element K block is multiplied as W_hi A_hi + W_hi A_lo + W_lo A_hi in ONE stage of the kernels (pf_conv_desc.split3:
    60 MFMAs on the 52 KB a plain stage moves for 40).  Rounds 2-3 walked K' = 3 K over [hi | lo] + [hi] against
    [W_hi | W_hi | W_lo]: the same three products, half again as many stages."""
    C2 = a_split.shape[-1]
    k = kw.get("ksize", 1)
    return ops.conv_gemm(a_split, w3, n_out, c0=C2, a0_ld=C2, algo_k=k * k * (C2 // 2), split3=True, **kw)


def to16(x, dtype):
    """Stream tensor -> 16-bit MFMA operand (no-op when the stream is 16-bit already)."""
    if x.dtype == dtype:
        return x
    c = x.shape[-1]
    return ops.scale_shift_act(x, None, 1, x.numel() // c, None, None, 0, out_dtype=dtype).view(x.shape)


def _norm(n, dev):
    return NS(g=_f32(n.weight, dev), b=_f32(n.bias, dev), eps=float(n.eps),
              groups=getattr(n, "num_groups", None))


def _bias(m, dev):
    return None if m.bias is None else _f32(m.bias, dev)


def pack_resnet(res, dev, dtype, mixed=False):
    r = NS()
    r.dtype, r.stream = dtype, (torch.float32 if mixed else dtype)
    r.cin, r.cout = res.conv1.weight.shape[1], res.conv1.weight.shape[0]
    r.norm1, r.norm2 = _norm(res.norm1, dev), _norm(res.norm2, dev)
    r.w1, r.b1 = _conv3_weight(res.conv1, dev, dtype), _bias(res.conv1, dev)
    r.w2, r.b2 = _conv3_weight(res.conv2, dev, dtype), _bias(res.conv2, dev)
    sc = getattr(res, "conv_shortcut", None)
    r.ws3 = None
    if sc is not None:
        r.ws = _w16(sc.weight.detach().float().reshape(r.cout, r.cin), dev, dtype)
        r.bs = _bias(sc, dev)
        if mixed:
            r.ws3 = _split_weight(sc.weight.detach().float().reshape(r.cout, r.cin), 1, dev, dtype)
    else:
        r.ws = r.bs = None
    r.temb = getattr(res, "time_emb_proj", None)      # consumed by pack_unet (concatenated projection); None: VAE resnets
    r.src = res                                       # (train_engine builds the backward operands from the module)
    return r


def pack_resnet_plain(res, dev, dtype, mixed=False):
    """A resnet without time embedding (the VAE decoder's, vae.py)."""
    r = pack_resnet(res, dev, dtype, mixed)
    del r.temb
    return r


def run_resnet_plain(r, x):
    return run_resnet(r, x, None, None)


def _current_lora(attn, name, lin):
    """The LoRA layer of one projection (PanoGenerator.py:132-151, rank-4 LoRA with scale 1), or None.  The LoRA matrices live
    in ONE of two places:
      * ``lin.lora_layer.{down,up}`` -- where diffusers 0.24 keeps them after the first Attention.forward migrated them out
        of the processor (and where its own saved checkpoints have them);
      * ``attn.processor.to_{q,k,v,out}_lora.{down,up}`` -- where ``unet.set_attn_processor(LoRAAttnProcessor(...))`` puts
        them and where a reference checkpoint loads them (convert_state_dict, PanoGenerator.py:101-111).  This engine never
        calls the modules' forward, so the migration never happens: both places are read here."""
    own = getattr(lin, "lora_layer", None)
    proc = _processor_lora(attn, name + "_lora")
    if own is not None and proc is not None:
        raise ValueError("LoRA weights found both in <linear>.lora_layer and in the attention processor")
    return own if own is not None else proc


def _dev32(t, dev):
    """fp32, contiguous, on the device -- the tensor itself when it already is (no copy of a frozen GPU weight)."""
    t = t.detach()
    if t.dtype == torch.float32 and t.device == torch.device(dev) and t.is_contiguous():
        return t
    return t.to(device=dev, dtype=torch.float32).contiguous()


def pack_attention(attn, dev, dtype, self_attn):
    """16-bit operands of one attention in PERSISTENT buffers: self-attention (q | k | v) rows of one [3C, C] tensor (the forward
    reads its (q | k) prefix and its v rows, the backward all of it), cross-attention q and (k | v), the output projection.
    ``a.proj[name]`` records where each projection's folded weight W + up @ down goes; fold_attention (re)writes them IN PLACE
    -- after an optimizer step on the LoRA matrices (MultiViewBaseModel.refold_lora) nothing is re-allocated, captured
    graphs and the backward's operands keep their addresses."""
    a = NS()
    a.heads, a.src, a.dtype, a.self_attn = attn.heads, attn, dtype, self_attn
    Cc, Dk = attn.to_q.weight.shape[0], attn.to_k.weight.shape[1]
    a.dim = Cc
    buf = lambda rows, cols: torch.empty(rows, cols, device=dev, dtype=dtype)
    if self_attn:
        a.wqkv = buf(3 * Cc, Cc)
        a.wqk, a.wv = a.wqkv[:2 * Cc], a.wqkv[2 * Cc:]
        dst = {"to_q": a.wqkv[:Cc], "to_k": a.wqkv[Cc:2 * Cc], "to_v": a.wv}
    else:
        a.wq, a.wkv = buf(Cc, Cc), buf(2 * Cc, Dk)
        a.wk, a.wv = a.wkv[:Cc], a.wkv[Cc:]
        dst = {"to_q": a.wq, "to_k": a.wk, "to_v": a.wv}
    a.wo = buf(Cc, Cc)
    dst["to_out"] = a.wo
    a.proj = {}
    for name, d in dst.items():
        lin = attn.to_out[0] if name == "to_out" else getattr(attn, name)
        # (w32: the frozen fp32 weight on the device -- the parameter itself when the module lives there)
        a.proj[name] = NS(lin=lin, dst=d, w32=_dev32(lin.weight, dev), dst_t=None, d_dst=None, u_dst=None)
    a.bo = _bias(attn.to_out[0], dev)
    fold_attention(a, everything=True)
    return a


def fold_attention(a, everything=False):
    """Write W + scale * up @ down of every LoRA-carrying projection of the packed attention `a` into its buffers (one
    pf_lora_fold launch each: folded weight, and -- once train_engine attached them -- its transpose and the 16-bit copies of
    the LoRA matrices the gradient GEMMs read).  everything: also the projections without LoRA (first packing / newly attached
    backward buffers); a re-fold after an optimizer step skips them, they cannot have changed."""
    dev = a.wo.device
    for name, pr in a.proj.items():
        lora = _current_lora(a.src, name, pr.lin)
        if lora is None:
            if everything:
                ops.lora_fold(pr.w32, None, None, 0.0, pr.dst, out_t=pr.dst_t)
            continue
        alpha = getattr(lora, "network_alpha", None)
        rank = lora.down.weight.shape[0]
        ops.lora_fold(pr.w32, _dev32(lora.up.weight, dev), _dev32(lora.down.weight, dev),
                      1.0 if alpha is None else float(alpha) / rank, pr.dst, out_t=pr.dst_t, d_out=pr.d_dst, u_out=pr.u_dst)


def pack_transformer(tf, dev, dtype, mixed=False):
    t = NS()
    t.src = tf
    t.dtype, t.stream = dtype, (torch.float32 if mixed else dtype)
    t.w_in3 = t.w_out3 = None
    if mixed:
        t.w_in3 = _split_weight(tf.proj_in.weight, 1, dev, dtype)
        t.w_out3 = _split_weight(tf.proj_out.weight, 1, dev, dtype)
    blk = tf.transformer_blocks[0]
    t.norm = _norm(tf.norm, dev)
    t.w_in, t.b_in = _w16(tf.proj_in.weight, dev, dtype), _bias(tf.proj_in, dev)
    t.w_out, t.b_out = _w16(tf.proj_out.weight, dev, dtype), _bias(tf.proj_out, dev)
    t.ln1, t.ln2, t.ln3 = _norm(blk.norm1, dev), _norm(blk.norm2, dev), _norm(blk.norm3, dev)
    t.attn1 = pack_attention(blk.attn1, dev, dtype, True)
    t.attn2 = pack_attention(blk.attn2, dev, dtype, False)
    # GEGLU.proj rows interleaved (value_j, gate_j): the gating runs in the GEMM epilogue
    t.w_ff1, t.b_ff1 = ops.interleave_geglu(_w16(blk.ff.net[0].proj.weight, dev, dtype), _bias(blk.ff.net[0].proj, dev))
    t.w_ff2, t.b_ff2 = _w16(blk.ff.net[2].weight, dev, dtype), _bias(blk.ff.net[2], dev)
    return t


def pack_unet(unet, dev, dtype, mixed=False):
    """Walks the diffusers attribute tree exactly as MVGenModel.py does."""
    u = NS()
    u.src = unet                               # (train_engine reaches the trainable ControlNet's parameters through it)
    u.dtype, u.mixed = dtype, mixed
    u.stream = torch.float32 if mixed else dtype
    ci = unet.conv_in
    u.cin, u.c0 = ci.weight.shape[1], ci.weight.shape[0]
    u.w_conv_in = _f32(ci.weight.detach().permute(2, 3, 1, 0), dev)           # [3,3,cin,cout]
    u.b_conv_in = _bias(ci, dev)
    co = getattr(unet, "conv_out", None)       # a ControlNet has the encoder half only
    if co is not None:
        u.cout = co.weight.shape[0]
        u.src_conv_out = co
        u.w_conv_out = _f32(co.weight.detach().permute(0, 2, 3, 1), dev)      # [cout,3,3,cin]
        # [3,3,cin,4]: the fused head kernel's layout (ops.conv_out_gn)
        u.w_conv_out_t = ops.conv_out_weight_t(co.weight).to(dev) if u.cout <= 4 and co.weight.shape[1] % 32 == 0 else None
        u.b_conv_out = _bias(co, dev)
        u.norm_out = _norm(unet.conv_norm_out, dev)
    te = unet.time_embedding
    u.t_dim = te.linear_1.weight.shape[1]
    u.w_t1, u.b_t1 = _w16(te.linear_1.weight, dev, dtype), _bias(te.linear_1, dev)
    u.w_t2, u.b_t2 = _w16(te.linear_2.weight, dev, dtype), _bias(te.linear_2, dev)

    resnets = []

    def res(r):
        p = pack_resnet(r, dev, dtype, mixed)
        resnets.append(p)
        return p

    u.down = []
    for blk in unet.down_blocks:
        b = NS(resnets=[res(r) for r in blk.resnets], attns=None, down=None)
        if getattr(blk, "has_cross_attention", False):
            b.attns = [pack_transformer(a, dev, dtype, mixed) for a in blk.attentions]
        if blk.downsamplers is not None:
            c = blk.downsamplers[0].conv
            b.down = NS(w=_conv3_weight(c, dev, dtype), b=_bias(c, dev), c=c.weight.shape[0], w3=None, src=c)
            if mixed:
                b.down.w3 = _split_weight(c.weight.detach().float().permute(0, 2, 3, 1).reshape(c.weight.shape[0], -1),
                                          9, dev, dtype)
        u.down.append(b)
    mid = unet.mid_block
    u.mid = NS(resnets=[res(r) for r in mid.resnets],
               attns=[pack_transformer(a, dev, dtype, mixed) for a in mid.attentions])
    u.up = []
    for blk in getattr(unet, "up_blocks", []):
        b = NS(resnets=[res(r) for r in blk.resnets], attns=None, up=None)
        if getattr(blk, "has_cross_attention", False):
            b.attns = [pack_transformer(a, dev, dtype, mixed) for a in blk.attentions]
        if blk.upsamplers is not None:
            c = blk.upsamplers[0].conv
            b.up = NS(w=_conv3_weight(c, dev, dtype), b=_bias(c, dev), c=c.weight.shape[0], src=c,
                      w4=_subpixel_weight(c, dev, dtype) if SUBPIXEL_UP else None, w4s=None)
            # mixed scheme: the LAST upsampling convolution (32^2 -> 64^2, level 0) maps the stream in split precision -- tools/precision_study.py
            # (S1+sp+d+hd+u@0, round 6): 10 % of the views' error and 9 % of the panorama's sit in this one layer's operand / weight rounding
            # (the other two: < 1 %); with the sub-pixel form's 2.25x fewer MACs the three products still cost less than the plain 9-tap conv did
            if mixed and SUBPIXEL_UP and EXACT_UP0 and len(u.up) == len(unet.up_blocks) - 2 and c.weight.shape[1] % 32 == 0:
                b.up.w4s = _split_weight(_subpixel_weight(c, dev, torch.float32), 4, dev, dtype)
        u.up.append(b)

    # one GEMM for every resnet's Linear(silu(temb)) (diffusers ResnetBlock2D.time_emb_proj)
    off = 0
    ws, bs = [], []
    for p in resnets:
        p.temb_off = off
        ws.append(p.temb.weight.detach().float())
        bs.append(p.temb.bias.detach().float())
        off += p.cout
        del p.temb
    u.w_temb = _w16(torch.cat(ws, 0), dev, dtype)
    u.b_temb = _f32(torch.cat(bs, 0), dev)
    u.temb_total = off
    return u


def _pad64(c):
    return (c + 63) // 64 * 64


def pack_controlnet(cn, dev, dtype, mixed=False):
    """diffusers ControlNetModel (PanoGenerator.py:153-157): the encoder half goes through pack_unet;
    the conditioning embedding (3 -> 16 -> 16 -> 32 -> 32 -> 96 -> 96 -> 256 -> c0, 3x3, three stride-2
    steps) runs on the same MFMA conv kernel with channel counts zero-padded to multiples of 64
    (activation buffers keep the padded stride, the pad columns stay zero); the 13 zero-convs are 1x1."""
    c = pack_unet(cn, dev, dtype, mixed)
    ce = cn.controlnet_cond_embedding
    first = ce.conv_in
    co0 = first.weight.shape[0]
    w0 = torch.zeros(3, 3, first.weight.shape[1], _pad64(co0))
    w0[..., :co0] = first.weight.detach().float().permute(2, 3, 1, 0)
    b0 = torch.zeros(_pad64(co0))
    b0[:co0] = first.bias.detach().float()
    c.ce_w0, c.ce_b0, c.ce_c0 = _f32(w0, dev), _f32(b0, dev), _pad64(co0)
    c.ce_layers = []
    for conv in [*ce.blocks, ce.conv_out]:
        co, ci = conv.weight.shape[:2]
        w = torch.zeros(co, 3, 3, _pad64(ci))
        w[..., :ci] = conv.weight.detach().float().permute(0, 2, 3, 1)
        c.ce_layers.append(NS(w=_w16(w.reshape(co, -1), dev, dtype), b=_bias(conv, dev), cout=co,
                              cin_pad=_pad64(ci), stride=conv.stride[0]))
    c.ce_bufs = {}
    zc = lambda m: NS(w=_w16(m.weight.detach().float().reshape(m.weight.shape[0], -1), dev, dtype), b=_bias(m, dev),
                      c=m.weight.shape[0])
    c.zero_down = [zc(m) for m in cn.controlnet_down_blocks]
    c.zero_mid = zc(cn.controlnet_mid_block)
    return c


def run_cond_embedding(c, cond):
    """cond (n, 3, H, W) -> NHWC [n, H/8, W/8, c0] (diffusers ControlNetConditioningEmbedding.forward)."""
    n, _, H, W = cond.shape
    dev = cond.device
    key = (n, H, W, str(dev))
    bufs = c.ce_bufs.get(key)
    if bufs is None:                       # persistent, zero-initialised: the padded channels are never written
        bufs = [torch.zeros(n, H, W, c.ce_c0, device=dev, dtype=c.dtype)]
        h, w = H, W
        for L in c.ce_layers:
            h, w = (h - 1) // L.stride + 1, (w - 1) // L.stride + 1
            bufs.append(torch.zeros(n, h, w, _pad64(L.cout), device=dev, dtype=c.dtype))
        c.ce_bufs[key] = bufs
    x = ops.conv_in(cond.float(), c.ce_w0, c.ce_b0, c.ce_c0, c.dtype, wrap=False, out=bufs[0])
    x = ops.silu(x, out=x)
    h, w = H, W
    for i, L in enumerate(c.ce_layers):
        out = bufs[i + 1]
        ops.conv_gemm(x, L.w, L.cout, n_img=n, h_in=h, w_in=w, ksize=3, stride=L.stride, pad=1, bias=L.b,
                      c0=L.cin_pad, out=out.view(-1, out.shape[-1]))
        if i + 1 < len(c.ce_layers):
            out = ops.silu(out, out=out)
        x, h, w = out, out.shape[1], out.shape[2]
    return x


def run_controlnet(c, latent, timestep, text, cond, make_branch=None, embed=None, rec=None):
    """ControlNetModel.forward(sample, timestep, encoder_hidden_states, controlnet_cond) on NHWC:
    -> (12 skip residuals [n, h, w, C], mid residual).  Plain zero-padded convolutions (the reference
    calls it on the un-padded panorama latent, MVGenModel.py:76-83).
    make_branch / embed / rec: the training forward (train_engine.controlnet_forward) substitutes a taping branch and a
    conditioning embedding that keeps its pre-activations, and gets the zero-convs' inputs back in rec."""
    br = (make_branch or Branch)(c, latent, timestep, text, pano=False, pad=False)
    br.h = ops.add(br.h, (embed or run_cond_embedding)(c, cond))
    br.skips = [br.h]
    for blk in c.down:
        for j, r in enumerate(blk.resnets):
            br.resnet(r)
            if blk.attns is not None:
                br.attention(blk.attns[j])
            br.push()
        if blk.down is not None:
            br.downsample(blk.down)
            br.push()
    br.resnet(c.mid.resnets[0])
    for a, r in zip(c.mid.attns, c.mid.resnets[1:]):
        br.attention(a)
        br.resnet(r)

    def zero_conv(z, x):
        n, h, w, Cc = x.shape
        return ops.conv_gemm(to16(x, c.dtype), z.w, z.c, n_img=n, h_in=h, w_in=w, ksize=1, bias=z.b).view(n, h, w, z.c)

    if rec is not None:
        rec.br, rec.skips, rec.h_mid = br, list(br.skips), br.h
    return [zero_conv(z, s) for z, s in zip(c.zero_down, br.skips)], zero_conv(c.zero_mid, br.h)


def pack_epa(block, dev, dtype, mixed=False):
    """block: module with the reference WarpAttn parameter names (modules.py:8-13)."""
    tr = block.transformer
    e = NS()
    e.cdtype, e.stream = dtype, (torch.float32 if mixed else dtype)
    e.dim = tr.norm1.weight.shape[0]
    e.heads = e.dim // 32
    e.ln1, e.ln2 = _norm(tr.norm1, dev), _norm(tr.norm2, dev)
    a = tr.attn1
    e.wqkv = _w16(torch.cat([a.to_q.weight.detach().float(), a.to_k.weight.detach().float(), a.to_v.weight.detach().float()], 0), dev, dtype)
    e.wqk, e.wv = e.wqkv[:2 * e.dim], e.wqkv[2 * e.dim:]
    e.wo, e.bo = _w16(a.to_out.weight, dev, dtype), _bias(a.to_out, dev)
    e.w_ff1, e.b_ff1 = ops.interleave_geglu(_w16(tr.ff.net[0].proj.weight, dev, dtype), _bias(tr.ff.net[0].proj, dev))
    e.w_ff2, e.b_ff2 = _w16(tr.ff.net[2].weight, dev, dtype), _bias(tr.ff.net[2], dev)
    e.freq = _f32(block.pe.freq_bands, dev)
    return e


# ---------------------------------------------------------------------------- layer runners
def run_resnet(r, x, skip, temb_all, groups_eps=None, wrap=0, save=None):
    """x [n, h, w, C] (+ skip concatenated along channels) -> [n, h, w, cout], stream dtype in and out.
    GN -> SiLU -> conv3x3 (+bias +temb) -> GN -> SiLU -> conv3x3 (+bias) + shortcut(x).
    GroupNorm moments: conv1's epilogue leaves the moments of h1 behind for norm2 (ops.conv_gemm(gn_stats=True)); norm1 uses
    what x / skip carry, else its own pass.
    wrap = p > 0: the panorama branch's pad_pano(x, p) -> resnet -> unpad_pano(., p) (MVGenModel.py:110-115) WITHOUT the padded
    copies, bug-compatible with the reference: norm1's statistics count the wrapped columns twice, conv1 reads the
    normalised x through a virtual circular padding and produces the w + 2p columns the reference has (its zero padding
    contaminating the two outermost ones), norm2 normalises exactly that tensor, conv2 produces only the w columns that
    survive the crop, and the 1x1 shortcut / identity acts on the un-padded x.
    save (a namespace, training forward): keeps what the backward reads -- both norms' (scale, shift) and h1, which is then
    produced in fp32 (the GroupNorm backward wants it at that precision) -- so that nothing is recomputed there."""
    n, h, w, _ = x.shape
    hw = h * w
    sc, sh = ops.groupnorm_scale_shift(x, skip, n, hw, r.norm1.groups, r.norm1.eps, r.norm1.g, r.norm1.b,
                                       wrap=(w, wrap) if wrap else None)
    if save is not None:
        save.sc1, save.sh1 = sc, sh
    pair = None
    if r.ws3 is not None and x.dtype == torch.float32 and RAW_PAIR_FUSED:
        # mixed scheme: the shortcut's split operand [hi | lo] of (x | skip) comes out of the same pass as norm1 + SiLU
        y, pair = ops.scale_shift_act(x, skip, n, hw, sc, sh, 1, out_dtype=r.dtype, raw_pair=True)
    else:
        y = ops.scale_shift_act(x, skip, n, hw, sc, sh, 1, out_dtype=r.dtype)
    rowvec = temb_all[:, r.temb_off:] if temb_all is not None else None
    wp = w + 2 * wrap
    h1 = ops.conv_gemm(y, r.w1, r.cout, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, bias=r.b1, rowvec=rowvec, gn_stats=True,
                       wrap_pad=wrap, out_dtype=torch.float32 if save is not None else None)
    h1 = ops.carry(h1.view(n, h * wp, r.cout), h1)
    sc, sh = ops.groupnorm_scale_shift(h1, None, n, h * wp, r.norm2.groups, r.norm2.eps, r.norm2.g, r.norm2.b)
    if save is not None:
        save.h1, save.sc2, save.sh2 = h1, sc, sh
    y2 = ops.scale_shift_act(h1, None, n, h * wp, sc, sh, 1, out_dtype=r.dtype)
    if r.ws3 is not None:         # mixed scheme: the shortcut maps the stream linearly -> split precision, fp32 out
        if pair is None:
            pair = split_operand(x, skip, dtype=r.dtype)
        short = exact_gemm(pair, r.ws3, r.cout, w_in=n * hw, bias=r.bs, out_dtype=r.stream)
    elif r.ws is not None:
        short = ops.conv_gemm(x, r.ws, r.cout, a1=skip, n_img=n, h_in=h, w_in=w, ksize=1, bias=r.bs)
    else:
        short = x.view(n * hw, r.cout)
    out = ops.conv_gemm(y2, r.w2, r.cout, n_img=n, h_in=h, w_in=wp, ksize=3, pad=1, bias=r.b2, residual=short, gn_stats=True,
                        crop=wrap)
    return ops.carry(out.view(n, h, w, r.cout), out)


def text_kv(a, text):
    """K and V^T of the text tokens for one cross-attention (attn2): text [n, L, Dt] -> (k [n*L, C], vt [n, C, ld])."""
    n, L = text.shape[:2]
    return ops.linear(text.reshape(n * L, -1), a.wk), ops.linear_t(text, a.wv)


def all_transformers(u):
    for blk in [*u.down, u.mid, *u.up]:
        for t in (blk.attns or []):
            yield t


def self_qkv(a, tokens, n, nk):
    """(q | k) [n * nk, 2C] and V^T [n, C, ld] of a self-attention from its layer-normed tokens: one launch where the shape allows
    (C = 320), else q | k + a transposed V projection."""
    fused = ops.linear_qkv(tokens, a.wqkv, n)
    if fused is not None:
        return fused
    qk = ops.linear(tokens, a.wqk)                            # [rows, 2C]  (q | k)
    vt = ops.linear_vt(tokens, a.wv, n)                       # weight-stationary kernel, transposed epilogue (C = 640 / 1280)
    if vt is None:
        vt = ops.linear_t(tokens.view(n, nk, -1), a.wv)       # [n, C, ld_v] keys contiguous
    return qk, vt


def _attend(a, q_src, kv_tokens, n, nq, nk, head_dim, *, self_attn, residual, kv=None, next_ln=None, split=None):
    """q_src [n*nq, C] already layer-normed; kv_tokens [n*nk, Ckv] (or kv = precomputed (k, vt)).
    next_ln: the LayerNorm that follows the output projection -> (tokens, its output or None): where the output projection runs on
    the weight-stationary kernel the norm rides in its epilogue (ops.linear_ln)."""
    Cq = a.dim
    vt = None
    if split is not None and self_attn and n == 1:
        # sharded panorama owner: the query rows of this self-attention are computed by all ranks of the CFG half (sharding.py):
        # the layer-normed tokens travel, every rank projects q | k | V^T itself
        from . import sharding
        o = sharding.split_pano_attention(split, a, q_src, nq)
        if next_ln is not None:
            return ops.linear_ln(o.view(n * nq, Cq), a.wo, a.bo, residual, next_ln.g, next_ln.b, next_ln.eps)
        return ops.linear(o.view(n * nq, Cq), a.wo, bias=a.bo, residual=residual)
    if self_attn:
        qk, vt = self_qkv(a, q_src, n, nk)
        q, k, ld = qk, qk[:, Cq:], 2 * Cq
    else:
        q, ld = ops.linear(q_src, a.wq), Cq
        k = kv[0] if kv is not None else ops.linear(kv_tokens, a.wk)
    if kv is not None:
        vt = kv[1]
    elif vt is None:
        vt = ops.linear_t(kv_tokens.view(n, nk, -1), a.wv)                # [n, C, ld_v] keys contiguous
    o = ops.attention(q, k, vt, n, a.heads, head_dim, nq, nk,
                      q_ld=ld, k_ld=(ld if self_attn else Cq), vt_ld=vt.shape[-1],
                      q_bs=nq * ld, k_bs=nk * (ld if self_attn else Cq), vt_bs=vt.shape[1] * vt.shape[2])
    if next_ln is not None:
        return ops.linear_ln(o.view(n * nq, Cq), a.wo, a.bo, residual, next_ln.g, next_ln.b, next_ln.eps)
    return ops.linear(o.view(n * nq, Cq), a.wo, bias=a.bo, residual=residual)


def run_transformer(t, x, text, kv=None, split=None):
    """diffusers Transformer2DModel (linear projections) on x [n, h, w, C] (stream dtype); text [n, L, Dt]
    (kv: the text K / V^T of this block computed ahead of time, text_kv)."""
    n, h, w, Cc = x.shape
    hw = h * w
    sc, sh = ops.groupnorm_scale_shift(x, None, n, hw, t.norm.groups, t.norm.eps, t.norm.g, t.norm.b)
    if t.w_in3 is not None:       # mixed scheme: proj_in / proj_out carry the stream -> split precision
        tok = exact_gemm(split_operand(x, None, sc, sh, 0, dtype=t.dtype), t.w_in3, Cc, w_in=n * hw, bias=t.b_in,
                         out_dtype=t.stream)
    else:
        y = ops.scale_shift_act(x, None, n, hw, sc, sh, 0)
        tok = ops.linear(y.view(n * hw, Cc), t.w_in, bias=t.b_in)
    dh = t.attn1.dim // t.attn1.heads
    ln = ops.layernorm(tok, t.ln1.g, t.ln1.b, t.ln1.eps, out_dtype=t.dtype)
    tok, ln = _attend(t.attn1, ln, ln, n, hw, hw, dh, self_attn=True, residual=tok, next_ln=t.ln2, split=split)
    if ln is None:
        ln = ops.layernorm(tok, t.ln2.g, t.ln2.b, t.ln2.eps, out_dtype=t.dtype)
    L = text.shape[1]
    tok, ln = _attend(t.attn2, ln, text.reshape(n * L, -1), n, hw, L, dh, self_attn=False, residual=tok, kv=kv, next_ln=t.ln3)
    if ln is None:
        ln = ops.layernorm(tok, t.ln3.g, t.ln3.b, t.ln3.eps, out_dtype=t.dtype)
    g = ops.linear(ln, t.w_ff1, bias=t.b_ff1, geglu=True)
    if t.w_out3 is not None and os.environ.get("PF_FF2_PAIR", "1") != "0":
        # the token stream's last value feeds proj_out only: FF2's epilogue emits it directly as the [hi | lo]
        # pair of the split-precision proj_out (no fp32 round trip, no separate split pass)
        pair = ops.linear(g, t.w_ff2, bias=t.b_ff2, residual=tok, split_out=True)
        out = exact_gemm(pair, t.w_out3, Cc, w_in=n * hw, bias=t.b_out, residual=x.view(n * hw, Cc), gn_stats=True)
    elif t.w_out3 is not None:           # A/B switch PF_FF2_PAIR=0: fp32 token stream + a separate split pass
        tok = ops.linear(g, t.w_ff2, bias=t.b_ff2, residual=tok)
        out = exact_gemm(split_operand(tok, dtype=t.dtype), t.w_out3, Cc, w_in=n * hw, bias=t.b_out, residual=x.view(n * hw, Cc),
                         gn_stats=True)
    else:
        tok = ops.linear(g, t.w_ff2, bias=t.b_ff2, residual=tok)
        out = ops.linear(tok, t.w_out, bias=t.b_out, residual=x.view(n * hw, Cc), gn_stats=True)
    return ops.carry(out.view(n, h, w, Cc), out)


class Branch:
    """One UNet driven layer by layer (cf. the per-branch statements of MVGenModel.py:85-297).
    ``pano=True`` wraps every conv-bearing module in circular width padding / cropping with
    the reference's pad and crop widths."""

    def __init__(self, u, latent, timestep, text, pano, pad):
        self.u, self.text, self.pad = u, text, (pano and pad)
        n = latent.shape[0]
        feats = ops.timestep_features(timestep, u.t_dim, u.dtype)              # [n, 320]
        e = ops.linear(feats, u.w_t1, bias=u.b_t1)
        e = ops.linear(ops.silu(e), u.w_t2, bias=u.b_t2)                       # emb [n, 1280]
        self.temb = ops.linear(ops.silu(e), u.w_temb, bias=u.b_temb, out_dtype=torch.float32)
        # pano: pad 1 / conv / crop 1 (MVGenModel.py:87-91) == circular-width convolution
        self.h = ops.conv_in(latent.float(), u.w_conv_in, u.b_conv_in, u.c0, u.stream, wrap=self.pad)
        self.skips = [self.h]
        self.text_kv = {}               # id(transformer pack) -> (k, vt) of the text tokens, if computed ahead
        self.text_ready = None          # event to wait for before the first use (computed on another stream)
        # sharded runs (MVGenModel sets them): the panorama owner's branch splits its big self-attentions over the CFG half
        # (attn_split = its ShardInfo), a view rank's branch helps right after its own (attn_help = callable(transformer pack))
        self.attn_split = None
        self.attn_help = None
        self.split_streams = None       # (main, side) when this branch runs on a side stream while its attentions are split
        self.keep = None                # the forward's list of tensors that cross streams (cleared at every join)

    def precompute_text_kv(self):
        """All 16 cross-attentions' K / V^T of the text tokens in one go: 32 tiny GEMMs that would otherwise sit
        between the big layers of the critical path.  They depend on the prompt only, not on the latents or the
        timestep: the result is kept with the packed weights and reused for as long as the SAME text tensor
        (storage + version counter) comes back -- every step of a sampling loop after the first."""
        key = (self.text.data_ptr(), self.text._version, tuple(self.text.shape), self.text.dtype, str(self.text.device))
        cache = self.u.__dict__.setdefault("text_kv_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 4:                       # a handful of prompts at most; entries a captured graph still reads are never evicted
                victim = next((ck for ck, h in cache.items() if not h.get("pins")), None)
                if victim is not None:
                    cache.pop(victim)
            hit = {id(t): text_kv(t.attn2, self.text) for t in all_transformers(self.u)}
            hit["text"] = self.text                   # keeps the storage (and so the key) alive
            cache[key] = hit
        self.text_kv = hit

    def _padded(self, t, p):
        return ops.pad_width(t, p) if (self.pad and t is not None) else t

    def resnet(self, r, skip=False, save=None):
        s = self.skips.pop() if skip else None
        if self.pad and VIRTUAL_PAD:
            self.h = run_resnet(r, self.h, s, self.temb, wrap=2, save=save)
        elif self.pad:
            out = run_resnet(r, self._padded(self.h, 2), self._padded(s, 2), self.temb, save=save)
            self.h = ops.crop_width(out, 2)
        else:
            self.h = run_resnet(r, self.h, s, self.temb, save=save)

    def attention(self, t):
        if self.text_ready is not None:
            torch.cuda.current_stream(self.h.device).wait_event(self.text_ready)
            self.text_ready = None
        split = None
        if self.attn_split is not None:
            from . import sharding
            if sharding.splits_pano_attention(self.attn_split, self.h.shape[1] * self.h.shape[2]) and self.h.shape[0] == 1:
                split = self.attn_split
        if split is not None and self.split_streams is not None:
            # The branch runs on a side stream (an owner that also holds views): the split's collectives break the hipGraph
            # segment (SegmentedGraph.eager ends the capture), which a forked side stream forbids -- and a collective issued from
            # the side stream would fall out of the one issue order every rank keeps.  Join, run this attention on the main
            # stream, fork again.  Tensors that cross the streams stay referenced until the next join.
            main, side = self.split_streams
            main.wait_stream(side)
            x_in = self.h
            with torch.cuda.stream(main):
                self.h = run_transformer(t, x_in, self.text, self.text_kv.get(id(t)), split=split)
            side.wait_stream(main)
            if self.keep is not None:
                self.keep.extend((x_in, self.h))
        else:
            self.h = run_transformer(t, self.h, self.text, self.text_kv.get(id(t)), split=split)
        if self.attn_help is not None:
            self.attn_help(t, self.h)

    def push(self):
        self.skips.append(self.h)

    def downsample(self, d):            # pano: pad 2, conv s2, crop 1   (MVGenModel.py:138-144)
        virt = self.pad and VIRTUAL_PAD
        x = self.h if virt else self._padded(self.h, 2)
        n, h, w, Cc = x.shape
        geo = dict(wrap_pad=2, crop=1) if virt else {}
        if d.w3 is not None:            # mixed scheme: stream -> stream linear map, split precision
            y = exact_gemm(split_operand(x, dtype=self.u.dtype), d.w3, d.c, n_img=n, h_in=h, w_in=w, ksize=3, stride=2,
                           pad=1, bias=d.b, out_dtype=self.u.stream, **geo)
        else:
            y = ops.conv_gemm(x, d.w, d.c, n_img=n, h_in=h, w_in=w, ksize=3, stride=2, pad=1, bias=d.b, **geo)
        if virt:
            self.h = y.view(n, (h - 1) // 2 + 1, y.shape[0] // (n * ((h - 1) // 2 + 1)), d.c)
            return
        y = y.view(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, d.c)
        self.h = ops.crop_width(y, 1) if self.pad else y

    def upsample(self, up):             # pano: pad 1, nearest x2 + conv, crop 2   (:272-277)
        virt = self.pad and VIRTUAL_PAD
        if getattr(up, "w4s", None) is not None and (virt or not self.pad) and self.h.dtype == torch.float32:
            n, h, w, Cc = self.h.shape                # split precision: the pair of the fp32 stream against [W_hi | W_lo] of the phase weights
            y = exact_gemm(split_operand(self.h, dtype=self.u.dtype), up.w4s, up.c, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, upsample=1,
                           bias=up.b, out_dtype=self.u.stream, gn_stats=not self.pad, subpixel=True, **(dict(wrap_pad=1, crop=2) if virt else {}))
            self.h = y.view(n, 2 * h, 2 * w, up.c) if virt else ops.carry(y.view(n, 2 * h, 2 * w, up.c), y)
            return
        x = to16(self.h if virt else self._padded(self.h, 1), self.u.dtype)
        n, h, w, Cc = x.shape
        geo = dict(wrap_pad=1, crop=2) if virt else {}
        sub = getattr(up, "w4", None) is not None and (virt or not self.pad)
        y = ops.conv_gemm(x, up.w4 if sub else up.w, up.c, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, upsample=1, bias=up.b,
                          out_dtype=self.u.stream, gn_stats=not self.pad, subpixel=sub, **geo)
        if virt:
            self.h = y.view(n, 2 * h, 2 * w, up.c)
            return
        y = ops.carry(y.view(n, 2 * h, 2 * w, up.c), y)
        self.h = ops.crop_width(y, 2) if self.pad else y

    def head(self):                     # GN + SiLU un-padded; conv_out padded by 1   (:279-294)
        u = self.u
        n, h, w, Cc = self.h.shape
        sc, sh = ops.groupnorm_scale_shift(self.h, None, n, h * w, u.norm_out.groups, u.norm_out.eps,
                                           u.norm_out.g, u.norm_out.b)
        # (mixed scheme: the 4-channel output conv reads the normalised activation in fp32 -- it is not an MFMA GEMM)
        if FUSED_HEAD and u.w_conv_out_t is not None and self.h.dtype == torch.float32 and self.h.is_contiguous():
            # GroupNorm-apply + SiLU inside the conv's LDS tile: one launch, the activated tensor never reaches HBM
            return ops.conv_out_gn(self.h, sc, sh, 1, u.w_conv_out_t, u.b_conv_out, u.cout, wrap=self.pad)
        y = ops.scale_shift_act(self.h, None, n, h * w, sc, sh, 1, out_dtype=u.stream).view(n, h, w, Cc)
        return ops.conv_out(y, u.w_conv_out, u.b_conv_out, u.cout, wrap=self.pad)      # fp32 NCHW


class EPATables:
    """Geometry that depends only on (cameras, sizes): bias tables, tile flags, PE tables.
    Built once per key on the device and kept (immutable keyed caches)."""

    MAX_ENTRIES = 16        # LRU bound: a rotation step that does not divide 360 would otherwise grow ~0.35 GB per step
    PINNING = 0             # > 0 while a hipGraph is being warmed up / captured: a captured graph reads its tables by
                            # address, so the sets fetched then are pinned and never evicted

    class pinned:
        def __enter__(self):
            EPATables.PINNING += 1

        def __exit__(self, *exc):
            EPATables.PINNING -= 1
            return False

    def __init__(self):
        self.cache = {}
        self.pins = set()

    def get(self, fov, theta, phi, ph, pw, eh, ew, freq, device):
        key = (tuple(float(v) for v in fov), tuple(float(v) for v in theta),
               tuple(float(v) for v in phi), ph, pw, eh, ew, freq.numel(), freq.data_ptr())
        if EPATables.PINNING:
            self.pins.add(key)
        hit = self.cache.pop(key, None)
        if hit is not None:
            self.cache[key] = hit               # most recently used last
        else:
            victims = [k for k in self.cache if k not in self.pins]
            while len(self.cache) >= self.MAX_ENTRIES and victims:
                self.cache.pop(victims.pop(0))
            bias_e, bias_p, flags_e, flags_p = ops.epa_tables(fov, theta, phi, ph, pw, eh, ew, device)
            _, _, lonlat = ops.e2p_grid(fov, theta, phi, eh, ew, ph, pw, device, want_lonlat=True)
            pe_p = ops.spherical_pe(lonlat.view(-1, 2), freq)              # [m*P, C]
            pe_e = ops.spherical_pe(ops.equi_coords(eh, ew, device).view(-1, 2), freq)   # [E, C]
            hit = NS(bias_e=bias_e, bias_p=bias_p, flags_e=flags_e, flags_p=flags_p, pe_p=pe_p, pe_e=pe_e)
            self.cache[key] = hit
        return hit


def _epa_tail(e, attn_out, x, Cc):
    """to_out + residual, then LN2 -> GEGLU FF -> + residual (transformer.py:159-161)."""
    y, ln2 = ops.linear_ln(attn_out.view(-1, Cc), e.wo, e.bo, x, e.ln2.g, e.ln2.b, e.ln2.eps)
    if ln2 is None:
        ln2 = ops.layernorm(y, e.ln2.g, e.ln2.b, e.ln2.eps, out_dtype=e.cdtype)
    g = ops.linear(ln2, e.w_ff1, bias=e.b_ff1, geglu=True)
    return ops.linear(g, e.w_ff2, bias=e.b_ff2, residual=y, gn_stats=True)


def run_epa_sharded(e, t, xp, xe, m, shard, equi_hw=None, pers_hw=None, posted=None):
    """EPA when this rank holds views [v0, v1) of the m (one CFG sample per rank, b == 1).
    One all-gather of LN1(x_p + PE) inside the CFG half; K / V^T of all views are projected locally.
    Replicated layout: every rank holds the panorama and computes the panorama-query direction redundantly.
    Panorama-rank layout (shard.pano_g): only the owner has xe; it computes the panorama-query direction and
    broadcasts LN1(x_e + PE), from which every rank projects the panorama K / V^T for its own view queries;
    the other ranks pass xe = None (equi_hw = its spatial size) and get None back for it.  An owner without
    views (explicit split 0, ...) passes xp = None and pers_hw = (ph, pw): it contributes an empty block to the
    gather, computes the panorama-query direction only and gets None back for the views.
    The two collectives are POSTED (sharding.ASYNC: async_op on the backend's communication stream) and awaited in front of their
    consumers: the owner layer-norms the panorama tokens while the view tokens travel, a view rank projects its own queries while
    the panorama tokens travel and never waits for the gathered view tokens it does not read.  posted: the all-gather handle a
    panorama-only owner posted before it ran the level's panorama resnets (WarpAttn.post_view_gather)."""
    from . import sharding
    owner = xe is not None
    if xp is None:
        if not owner:
            raise ValueError("a rank without views must own the panorama branch")
        (ph, pw), mloc, Cc = pers_hw, 0, xe.shape[-1]
    else:
        mloc, ph, pw, Cc = xp.shape
    if owner:
        b, eh, ew, _ = xe.shape
        if b != 1:
            raise ValueError("sharded EPA expects one CFG sample per rank")
    else:
        eh, ew = equi_hw
    P, E = ph * pw, eh * ew
    mP = m * P
    v0, v1 = shard.views
    r0, r1 = v0 * P, v1 * P
    flags_p_loc = t.flags_p[r0 // 32:(r1 + 31) // 32]
    if r0 % 32:
        # view-group boundary inside a 32-row flag tile (only at toy sizes, P < 32): re-derive the tile
        # map of the local rows from the table itself (one-off table preparation, not step arithmetic)
        key = ("flags_p_loc", r0, r1)
        if key not in t.__dict__:
            rows = t.bias_p[r0:r1]
            pad = torch.nn.functional.pad(rows, (0, (-E) % 32, 0, (-(r1 - r0)) % 32))
            t.__dict__[key] = (pad.reshape(pad.shape[0] // 32, 32, pad.shape[1] // 32, 32).abs().amax((1, 3)) > 0) \
                .to(torch.uint8).contiguous()
        flags_p_loc = t.__dict__[key]
    if mloc:
        tp = xp.view(mloc * P, Cc)
        lnp_loc = ops.layernorm(tp, e.ln1.g, e.ln1.b, e.ln1.eps, pe=t.pe_p[r0:r1], out_dtype=e.cdtype)
    else:
        tp = xe.new_empty(0, Cc)
        lnp_loc = torch.empty(0, Cc, device=xe.device, dtype=e.cdtype)
    if posted is not None:
        gathered = posted                                                   # (a rank without views: its block of the gather is empty)
    elif sharding.ASYNC:
        gathered = sharding.gather_view_tokens_async(lnp_loc, shard, P=P)   # posted; every rank takes part
    else:
        gathered = sharding._Ready(sharding.gather_view_tokens(lnp_loc, shard, P=P))     # [mP, C]
    lne = None
    if owner:
        te = xe.view(E, Cc)
        lne = ops.layernorm(te, e.ln1.g, e.ln1.b, e.ln1.eps, pe=t.pe_e, out_dtype=e.cdtype)
    if sharding.ASYNC:
        shared = sharding.share_pano_tokens_async(lne, E, Cc, lnp_loc, shard)        # broadcast in the panorama-rank layout
    else:
        shared = sharding._Ready(sharding.share_pano_tokens(lne, E, Cc, lnp_loc, shard))
    q_loc = None
    if not owner:
        q_loc = ops.linear(lnp_loc, e.wqk)                                  # only the local queries are needed: while the tokens travel
    lne = shared.result()
    qk_e = ops.linear(lne, e.wqk)
    vt_e = ops.linear_t(lne.view(1, E, Cc), e.wv)
    ld = 2 * Cc
    out_e = None
    if owner:
        lnp = gathered.result()
        qk_p = ops.linear(lnp, e.wqk)
        vt_p = ops.linear_vt(lnp, e.wv, 1)
        if vt_p is None:
            vt_p = ops.linear_t(lnp.view(1, mP, Cc), e.wv)
        a_e = ops.attention(qk_e, qk_p[:, Cc:], vt_p, 1, e.heads, 32, E, mP, q_ld=ld, k_ld=ld, vt_ld=vt_p.shape[-1],
                            q_bs=E * ld, k_bs=mP * ld, vt_bs=vt_p.shape[1] * vt_p.shape[2],
                            bias=t.bias_e, flags=t.flags_e)
        out_e = _epa_tail(e, a_e, te, Cc).view(1, eh, ew, Cc)
        if not mloc:
            return None, out_e
        q_loc = qk_p[r0:r1]
    nq = mloc * P
    a_p = ops.attention(q_loc, qk_e[:, Cc:], vt_e, 1, e.heads, 32, nq, E, q_ld=ld, k_ld=ld,
                        vt_ld=vt_e.shape[-1], q_bs=nq * ld, k_bs=E * ld, vt_bs=vt_e.shape[1] * vt_e.shape[2],
                        bias=t.bias_p[r0:r1], flags=flags_p_loc)
    out_p = _epa_tail(e, a_p, tp, Cc)
    if not owner and hasattr(gathered, "finish"):
        gathered.finish()                                                   # (the all-gather this rank only contributed to)
    return out_p.view(mloc, ph, pw, Cc), out_e


def run_epa(e, tables, xp, xe, m, shard=None, equi_hw=None, side=None, pers_hw=None, n_samples=None, posted=None):
    """EPA fusion (modules.py:15-59) on NHWC activations.
    xp [b*m, ph, pw, C], xe [b, eh, ew, C]; tables: list with one EPATables entry per batch
    element (or a single shared entry)."""
    if shard is not None:
        b = n_samples or 1
        if b == 1:
            if len(tables) != 1:
                raise ValueError("sharded EPA needs one camera set per sample")
            return run_epa_sharded(e, tables[0], xp, xe, m, shard, equi_hw, pers_hw, posted=posted)
        # a multi-prompt batch on a sharded rank (b samples of the rank's CFG half): sample by sample -- every rank walks the samples
        # in the same order, so the collectives of the b passes line up
        mloc = xp.shape[0] // b if xp is not None else 0
        outs_p, outs_e = [], []
        for i in range(b):
            t = tables[0] if len(tables) == 1 else tables[i]
            op, oe = run_epa_sharded(e, t, xp[i * mloc:(i + 1) * mloc] if xp is not None else None,
                                     xe[i:i + 1] if xe is not None else None, m, shard, equi_hw, pers_hw)
            outs_p.append(op)
            outs_e.append(oe)
        return (torch.cat(outs_p) if outs_p[0] is not None else None), (torch.cat(outs_e) if outs_e[0] is not None else None)
    bm, ph, pw, Cc = xp.shape
    b, eh, ew, _ = xe.shape
    P, E = ph * pw, eh * ew
    mP = m * P
    tp, te = xp.view(b * mP, Cc), xe.view(b * E, Cc)
    shared = len(tables) == 1
    if shared:
        lnp = ops.layernorm(tp, e.ln1.g, e.ln1.b, e.ln1.eps, pe=tables[0].pe_p, out_dtype=e.cdtype)
        lne = ops.layernorm(te, e.ln1.g, e.ln1.b, e.ln1.eps, pe=tables[0].pe_e, out_dtype=e.cdtype)
    else:
        lnp, lne = torch.empty_like(tp, dtype=e.cdtype), torch.empty_like(te, dtype=e.cdtype)
        for i, t in enumerate(tables):
            ops.layernorm(tp[i * mP:(i + 1) * mP], e.ln1.g, e.ln1.b, e.ln1.eps, pe=t.pe_p, out=lnp[i * mP:(i + 1) * mP])
            ops.layernorm(te[i * E:(i + 1) * E], e.ln1.g, e.ln1.b, e.ln1.eps, pe=t.pe_e, out=lne[i * E:(i + 1) * E])
    fused_p, fused_e = ops.linear_qkv(lnp, e.wqkv, b), ops.linear_qkv(lne, e.wqkv, b)   # one launch for q | k | v where the shape allows
    qk_p, vt_p = fused_p if fused_p is not None else (ops.linear(lnp, e.wqk), ops.linear_t(lnp.view(b, mP, Cc), e.wv))   # [., 2C] = (q | k), [b, C, mP]
    qk_e, vt_e = fused_e if fused_e is not None else (ops.linear(lne, e.wqk), ops.linear_t(lne.view(b, E, Cc), e.wv))    # [b, C, E]
    ld = 2 * Cc

    def attend(q, k, vt, nq, nk, which):
        kw = dict(q_ld=ld, k_ld=ld, vt_ld=vt.shape[-1], q_bs=nq * ld, k_bs=nk * ld,
                  vt_bs=vt.shape[1] * vt.shape[2])
        if shared:
            t = tables[0]
            bias, flags = (t.bias_e, t.flags_e) if which == "e" else (t.bias_p, t.flags_p)
            return ops.attention(q, k, vt, b, e.heads, 32, nq, nk, bias=bias, flags=flags, **kw)
        out = torch.empty(b, nq, Cc, device=q.device, dtype=q.dtype)
        for i, t in enumerate(tables):
            bias, flags = (t.bias_e, t.flags_e) if which == "e" else (t.bias_p, t.flags_p)
            ops.attention(q[i * nq:], k[i * nk:], vt[i], 1, e.heads, 32, nq, nk, bias=bias, flags=flags,
                          out=out[i], **kw)
        return out

    def tail(attn_out, x):
        return _epa_tail(e, attn_out, x, Cc)

    # panorama pixels query the views (modules.py:43-48), then views query the panorama with the
    # ORIGINAL view activations (modules.py:50-55): the two directions are independent -- with a side stream
    # the (small, latency-bound) panorama side runs next to the view side
    if side is not None:
        main = torch.cuda.current_stream(xp.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out_e = tail(attend(qk_e, qk_p[:, Cc:], vt_p, E, mP, "e"), te)
        out_p = tail(attend(qk_p, qk_e[:, Cc:], vt_e, mP, E, "p"), tp)
        main.wait_stream(side)
    else:
        out_e = tail(attend(qk_e, qk_p[:, Cc:], vt_p, E, mP, "e"), te)
        out_p = tail(attend(qk_p, qk_e[:, Cc:], vt_e, mP, E, "p"), tp)
    return ops.carry(out_p.view(bm, ph, pw, Cc), out_p), ops.carry(out_e.view(b, eh, ew, Cc), out_e)
