"""Backward of the dual-branch denoiser: the training step of the reference through the same boundary
(``PanFusion.training_step``, PanFusion.py:64-98: one ``mv_base_model`` call, MSE on both outputs; trainable are the EPA
blocks and the rank-4 LoRA matrices on the attention projections of both UNets, PanoGenerator.py:129-160 -- the UNet
weights themselves are frozen).

The forward of a training step runs the inference engine (engine.Branch, same kernels, same precision scheme) through a
``TrainBranch`` that notes, per layer, which packed layer ran on which input tensors AND keeps what the layer's backward reads
(KEEP, round 3: the resnets' norms and h1, the transformers' and EPA blocks' whole records -- their training forward is the code
that used to be the backward's recompute; PF_TRAIN_KEEP=0 recomputes from the inputs instead).  The backward walks that tape
in reverse, the panorama branch on the side stream beside the view branch as in the forward (TWO_STREAMS), and stops at the
earliest entry with a trainable leaf behind it; every entry applies its layer's backward (16-bit MFMA operands, fp32
residual stream):

  resnet        GN+SiLU backward (pf_groupnorm_bwd), 3x3 data gradients = the forward GEMM kernel on flipped,
                transposed weights, 1x1 shortcut on the transposed weight, the two-source concat splits into (dx, dskip)
  transformer   GN / proj_in / [LN - self-attention - LN - text cross-attention - LN - GEGLU FF] / proj_out, attention
                backward from the recomputed log-sum-exp, LoRA gradients as four thin GEMMs per projection
  down / up     stride-2 conv: zero-stuffed gradient through the stride-1 kernel; nearest x2 + conv: 2x2 sums after it
  panorama      circular pad / crop around every conv-bearing module (MVGenModel.py:110-115): fold / zero-margin kernels
  head          conv_out's data gradient = the boundary conv kernel (conv_in) on rearranged weights, then GN+SiLU backward
  EPA fusion    training.epa_recompute / epa_backward
  ControlNet    (layout conditions, PanoGenerator.py:153-157: ALL of its parameters train) a tape of its own, walked by the same
                layer backward passes with a weight sink: conv / linear weight gradients (im2col + token-reducing GEMM), GroupNorm /
                LayerNorm parameters, time embedding, conditioning embedding, the 13 zero-convs (controlnet_backward)

Every entry normalises its incoming gradient by a power of two on the device (max |d| in [1, 2)) before it becomes a
16-bit operand and scales its results back (training.py: an MSE over 1e5 latent elements hands down 1e-5).
"""
import torch

from . import engine, ops, training

import os

NS = engine.NS
F32 = torch.float32
# The linear maps that carry the gradient STREAM onto itself (proj_out / proj_in, the 1x1 shortcut, the stride-2 conv: the
# transposes of the forward's stream-path GEMMs) run split-precision like their forward counterparts (engine.exact_gemm):
# their operand roundings accumulate along the identity path of the residual network.  PF_TRAIN_EXACT=0: single pass (A/B).
EXACT = os.environ.get("PF_TRAIN_EXACT", "1") != "0"
# The training forward KEEPS what the backward reads (288 GB of HBM: the whole step peaks at 25-45 GB) instead of recomputing
# every layer inside its backward as rounds 2 did (the reference checkpoints only the EPA blocks, transformer.py:77-127; the
# UNets' activations stay alive under autograd there too).  PF_TRAIN_KEEP=0: recompute (A/B, and the lower-memory mode).
KEEP = os.environ.get("PF_TRAIN_KEEP", "1") != "0"
# Forward AND backward of a training step run the panorama branch on the model's side stream next to the view branch, joined at
# the EPA blocks, as the inference step does (neither branch fills 256 CUs at the training resolution).  PF_TRAIN_STREAMS=0: one stream.
TWO_STREAMS = os.environ.get("PF_TRAIN_STREAMS", "1") != "0"


# ---------------------------------------------------------------------------------------------- weights of the backward GEMMs
def flip_conv3_weight(w, dtype):
    """torch conv weight [cout, cin, 3, 3] -> data-gradient operand [cin, 9 * cout]: Wd[ci][ky][kx][co] = W[co][ci][2-ky][2-kx]."""
    w = w.detach().float()
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).to(dtype).contiguous()


def conv_out_dgrad_weight(w):
    """conv_out weight [cout, C, 3, 3] -> the boundary conv kernel's layout [3, 3, cin = cout, C] for its data gradient."""
    return w.detach().float().flip(2, 3).permute(2, 3, 0, 1).contiguous()


def _t16(w, dev, dtype):
    """W [N, K] -> W^T [K, N] as a GEMM weight operand (the data gradient dx = dy W is a GEMM with weight W^T)."""
    return w.detach().to(device=dev, dtype=F32).t().to(dtype).contiguous()


def _w16(w, dev, dtype):
    return w.detach().to(device=dev, dtype=F32).to(dtype).contiguous()


def _t3(w, dev, dtype):
    """W [N, K] -> the split-precision packing of W^T (engine._split_weight) for a stream-path data gradient."""
    return engine._split_weight(w.detach().to(device=dev, dtype=F32).t().contiguous(), 1, dev, dtype)


def stream_linear(d, w_t, w_t3, n_out, dtype):
    """d fp32 [T, K] (a gradient stream) times W: split precision when the packed W^T triple exists, else one 16-bit pass."""
    if w_t3 is not None:
        return engine.exact_gemm(engine.split_operand(d, dtype=dtype), w_t3, n_out, w_in=d.shape[0], out_dtype=F32)
    return ops.linear(engine.to16(d, dtype), w_t, out_dtype=F32)


def conv3_dgrad(dy, wflip, cin, mode, dtype):
    """dy fp32 NHWC [n, ho, wo, cout] -> dx fp32 NHWC.  mode "s1": 3x3 pad 1; "s2": stride 2 (dx [n, 2ho, 2wo, cin]);
    "up": nearest x2 before the conv (dx [n, ho/2, wo/2, cin])."""
    n, ho, wo, cout = dy.shape
    d16 = engine.to16(dy, dtype)
    if mode == "s2":
        d16 = ops.zero_insert2(d16)
        ho, wo = 2 * ho, 2 * wo
    dx = ops.conv_gemm(d16, wflip, cin, n_img=n, h_in=ho, w_in=wo, ksize=3, pad=1, out_dtype=F32).view(n, ho, wo, cin)
    return ops.sum2x2(dx) if mode == "up" else dx


def conv_out_dgrad(d_eps, wgt, cin, wrap):
    """d_eps fp32 NCHW [n, cout, h, w] -> fp32 NHWC [n, h, w, cin] (circular in width for the panorama)."""
    zero = torch.zeros(cin, device=d_eps.device, dtype=F32)
    return ops.conv_in(d_eps.float(), wgt, zero, cin, F32, wrap=wrap)


def _normalise(d):
    state = ops.grad_scale_state([d])
    return ops.scale_by_state(d, state, 1), state


def _unscale(t, state):
    return ops.scale_by_state(t, state, 2, out=t)


# ---------------------------------------------------------------------------------------------- weight gradients
# The layers of a TRAINABLE network: the ControlNet under layout conditions (PanoGenerator.py:153-157 -- every ControlNet
# parameter trains at lr x 0.1; its forward is called at MVGenModel.py:68-83).  The UNets themselves stay frozen.
_WGRAD_CHUNK = 1 << 30        # bytes of im2col columns per weight-gradient GEMM (pf_conv_gemm addresses operands below 2 GiB)
_UNIT = {}


def _unit_affine(C, dev):
    key = (C, str(dev))
    hit = _UNIT.get(key)
    if hit is None:
        hit = _UNIT[key] = (torch.ones(C, device=dev, dtype=F32), torch.zeros(C, device=dev, dtype=F32))
    return hit


def conv3_wgrad(d16, x16, stride=1):
    """Weight gradient of a 3x3 / pad 1 convolution in torch layout [cout, cin, 3, 3]: d16 [n, ho, wo, cout] the 16-bit
    output gradient, x16 [n, h, w, cin] the 16-bit input.  im2col (pf_im2col3) + ONE token-reducing MFMA GEMM per chunk of
    images (training.weight_grad: dW = dY^T cols)."""
    n, ho, wo, cout = d16.shape
    cin = x16.shape[-1]
    step = max(1, min(n, _WGRAD_CHUNK // (ho * wo * 9 * cin * 2)))
    g = None
    for i in range(0, n, step):
        cols = ops.im2col3(x16[i:i + step], stride)
        part = training.weight_grad(d16[i:i + step].reshape(-1, cout), cols)
        g = part if g is None else ops.add(g, part)
    return g.view(cout, 3, 3, cin).permute(0, 3, 1, 2)


def gn_param_grads(norm, x, skip, n, hw, sc, sh, act, dy):
    """(dgamma, dbeta) of a GroupNorm (+ SiLU) whose affine forward was (sc, sh); norm: the packed norm (groups, eps)."""
    C = x.shape[-1] + (skip.shape[-1] if skip is not None else 0)
    ones, zeros = _unit_affine(C, x.device)
    usc, ush = ops.groupnorm_scale_shift(x, skip, n, hw, norm.groups, norm.eps, ones, zeros)
    return ops.groupnorm_param_grads(x, skip, n, hw, sc, sh, usc, ush, act, dy)


def _scaled(wsink, state):
    """wsink for gradients computed in a layer's normalised units: scales them back before they reach the parameter sink."""
    def put(param, grad):
        g = grad.to(F32).contiguous()
        wsink(param, _unscale(g, state))
    return put


# ---------------------------------------------------------------------------------------------- LoRA
class LoRARef:
    """One rank-r LoRA pair of a projection: W' = W + s * up @ down (engine.fold_attention folds it for the forward)."""

    def __init__(self, lora):
        self.down, self.up = lora.down.weight, lora.up.weight               # [r, K], [N, r]
        alpha = getattr(lora, "network_alpha", None)
        self.scale = 1.0 if alpha is None else float(alpha) / self.down.shape[0]


def lora_of(attn, lin, name):
    lora = getattr(lin, "lora_layer", None) or engine._processor_lora(attn, name)
    return None if lora is None else LoRARef(lora)


class LoRAGroup:
    """The LoRA pairs of projections that read the SAME input x (q / k / v of a self-attention; k / v of the text
    cross-attention; a single projection) and whose output gradients sit side by side in one [T, sum N_i] tensor.
    Their gradients come out of two thin GEMMs and two weighted column sums for the whole group:
        P^T = D X^T          D = the down matrices stacked                      [R, K] x [T, K]^T -> [R, T]   (MFMA GEMM, fp32 out)
        Q^T = U_bd dY^T      U_bd = the up^T matrices on a block diagonal      [R, N] x [T, N]^T -> [R, T]
        d_up   = dY^T P      sum_t P^T[r, t] dY[t, n]    (pf_weighted_colsum: dY read once, row-major; pair i: block i of [R, N])
        d_down = Q^T X       sum_t Q^T[r, t] X[t, k]     (pair i: rows of block i)
    with R = sum of the ranks (rounded up to 4).  (Round 2 ran the two reductions as MFMA GEMMs on transposed copies of X and dY:
    two transposes, two 4-row GEMMs with split-K reduces and four scaling / slicing passes per group -- ~160 groups per step.)
    D and U are allocated ONCE (zero outside the blocks); their blocks are (re)written by engine.fold_attention together with
    the folded weights -- `projs`: the packed attention's projection records, which get the views to write to."""

    def __init__(self, refs, widths, projs, dev, dtype):
        self.refs, self.widths = refs, widths
        self.live = [(i, r) for i, r in enumerate(refs) if r is not None]
        if not self.live:
            return
        K = self.live[0][1].down.shape[1]
        N = sum(widths)
        ranks = [r.down.shape[0] for _, r in self.live]
        self.R = (sum(ranks) + 3) // 4 * 4
        self.D = torch.zeros(self.R, K, device=dev, dtype=dtype)
        self.U = torch.zeros(self.R, N, device=dev, dtype=dtype)
        self.slots = []
        row = 0
        for (i, ref), rk in zip(self.live, ranks):
            col = sum(widths[:i])
            projs[i].d_dst = self.D[row:row + rk]
            projs[i].u_dst = self.U[row:row + rk, col:col + widths[i]]
            self.slots.append((ref, row, rk, col, widths[i]))
            row += rk

    def grads(self, x16, dy16, sink):
        """x16 [T, K] the projections' input, dy16 [T, N] the gradient of their outputs (16-bit, row-major as the backward holds
        them; T a multiple of 4); sink: the block's ScaledSink (its normalisation factor is applied inside the reduction)."""
        if not self.live or not _trains(self.refs):
            return                                    # (frozen LoRA: the reference's layout-conditioned runs, PanoGenerator.py:173)
        T = x16.shape[0]
        pt = ops.conv_gemm(self.D, x16, T, w_in=self.R, out_dtype=F32)                         # P^T [R, T]
        qt = ops.conv_gemm(self.U, dy16, T, w_in=self.R, out_dtype=F32)                        # Q^T [R, T]
        unscale = sink.state[2:3]                                                              # 2^e of the block's normalisation
        blocks = [(row, rk, col, width) for _, row, rk, col, width in self.slots]
        d_up = ops.weighted_colsum(dy16, pt, dev_scale=unscale, blocks=blocks)                 # per pair [N_i, rank], flat
        d_down = ops.weighted_colsum(x16, qt, dev_scale=unscale)                               # [R, K]
        off = 0
        for ref, row, rk, col, width in self.slots:
            sink.sink(ref.up, d_up[off:off + width * rk].view(width, rk), ref.scale)
            sink.sink(ref.down, d_down[row:row + rk], ref.scale)
            off += width * rk


# ---------------------------------------------------------------------------------------------- per-layer training packs
def resnet_train(r, dev):
    """Backward operands of a packed resnet (frozen weights: built once, kept on the pack)."""
    tw = getattr(r, "train", None)
    if tw is None:
        src = r.src
        tw = NS(w1=flip_conv3_weight(src.conv1.weight.to(dev), r.dtype), w2=flip_conv3_weight(src.conv2.weight.to(dev), r.dtype), ws=None)
        sc = getattr(src, "conv_shortcut", None)
        tw.ws3 = None
        if sc is not None:
            tw.ws = _t16(sc.weight.reshape(r.cout, r.cin), dev, r.dtype)            # [cin, cout]
            if EXACT:
                tw.ws3 = _t3(sc.weight.reshape(r.cout, r.cin), dev, r.dtype)
        r.train = tw
    return tw


def _attn_train(a, attn, dev, dtype, self_attn):
    """Forward + backward operands of one attention, built ONCE per inference pack `a` and kept on it: the folded weights are
    the pack's own buffers, their transposes (data-gradient operands) and the LoRA groups' stacked matrices are allocated here
    and registered with the pack's projection records -- from then on engine.fold_attention (MultiViewBaseModel.refold_lora, at
    the top of every forward) rewrites all of them in place when a LoRA matrix changed.  Rebuilt only when the set of
    attached LoRA modules changes."""
    names = ("to_q", "to_k", "to_v", "to_out")
    ref = {n: lora_of(attn, a.proj[n].lin, n + "_lora") for n in names}
    ids = tuple(0 if ref[n] is None else id(ref[n].down) for n in names)
    t = getattr(a, "train", None)
    if t is not None and t.lora_ids == ids:
        return t
    t = NS(lora_ids=ids, heads=a.heads, dim=a.dim)
    Cc = a.dim
    pr = a.proj
    for p_ in pr.values():
        p_.dst_t = p_.d_dst = p_.u_dst = None
    if self_attn:
        t.wqkv = a.wqkv                                                          # [3C, C] = (q | k | v)
        t.wqkv_t = torch.empty(Cc, 3 * Cc, device=dev, dtype=dtype)
        for i, n in enumerate(names[:3]):
            pr[n].dst_t = t.wqkv_t[:, i * Cc:(i + 1) * Cc]
        t.lora_qkv = LoRAGroup([ref["to_q"], ref["to_k"], ref["to_v"]], [Cc, Cc, Cc], [pr[n] for n in names[:3]], dev, dtype)
    else:
        t.wq, t.wq_t = a.wq, torch.empty(Cc, Cc, device=dev, dtype=dtype)
        pr["to_q"].dst_t = t.wq_t
        t.wkv = a.wkv
        t.lora_q = LoRAGroup([ref["to_q"]], [Cc], [pr["to_q"]], dev, dtype)
        t.lora_kv = LoRAGroup([ref["to_k"], ref["to_v"]], [Cc, Cc], [pr["to_k"], pr["to_v"]], dev, dtype)
    t.wv = a.wv
    t.wo, t.wo_t, t.bo = a.wo, torch.empty(Cc, Cc, device=dev, dtype=dtype), a.bo
    pr["to_out"].dst_t = t.wo_t
    t.lora_out = LoRAGroup([ref["to_out"]], [Cc], [pr["to_out"]], dev, dtype)
    engine.fold_attention(a, everything=True)                                    # fills the new buffers (and the weights again)
    a.train = t
    return t


def transformer_train(t, dev):
    """Backward operands of a packed transformer: the frozen parts once; the LoRA-carrying attentions' operands live on the
    inference pack and are refreshed in place with the folded weights (_attn_train)."""
    src = t.src
    blk = src.transformer_blocks[0]
    tw = getattr(t, "train", None)
    if tw is None:
        tw = NS()
        tw.w_in, tw.w_in_t = _w16(src.proj_in.weight.reshape(src.proj_in.weight.shape[0], -1), dev, t.dtype), \
            _t16(src.proj_in.weight.reshape(src.proj_in.weight.shape[0], -1), dev, t.dtype)
        tw.w_out_t = _t16(src.proj_out.weight.reshape(src.proj_out.weight.shape[0], -1), dev, t.dtype)
        tw.w_in_t3 = tw.w_out_t3 = None
        if EXACT:
            tw.w_in_t3 = _t3(src.proj_in.weight.reshape(src.proj_in.weight.shape[0], -1), dev, t.dtype)
            tw.w_out_t3 = _t3(src.proj_out.weight.reshape(src.proj_out.weight.shape[0], -1), dev, t.dtype)
        ff1, ff2 = blk.ff.net[0].proj, blk.ff.net[2]
        tw.w1, tw.w1_t, tw.b1 = _w16(ff1.weight, dev, t.dtype), _t16(ff1.weight, dev, t.dtype), engine._bias(ff1, dev)
        tw.w2, tw.w2_t = _w16(ff2.weight, dev, t.dtype), _t16(ff2.weight, dev, t.dtype)
        t.train = tw
    tw.attn1 = _attn_train(t.attn1, blk.attn1, dev, t.dtype, True)
    tw.attn2 = _attn_train(t.attn2, blk.attn2, dev, t.dtype, False)
    return tw


TEXT_PAD = 128        # text keys per sample in the recomputed cross-attention: 77 tokens + masked zero rows (token counts of
                      # the LoRA reductions must be multiples of 64, pf_attention_bwd wants whole key quads under a bias)
_TEXT_BIAS = {}


def text_bias(nq, L, dev):
    """Bias table [nq, TEXT_PAD] that masks the padded text keys (-1e30 on columns >= L) + its 32x32 tile flags."""
    key = (nq, L, str(dev))
    hit = _TEXT_BIAS.get(key)
    if hit is None:
        if len(_TEXT_BIAS) > 16:
            _TEXT_BIAS.clear()
        bias = torch.zeros(nq, TEXT_PAD, device=dev, dtype=F32)
        bias[:, L:] = -1e30
        flags = torch.zeros((nq + 31) // 32, TEXT_PAD // 32, device=dev, dtype=torch.uint8)
        flags[:, L // 32:] = 1
        hit = _TEXT_BIAS[key] = (bias, flags)
    return hit


def pad_text(text, dtype):
    """text [n, L, Dt] -> [n, TEXT_PAD, Dt] 16-bit with zero rows behind the L tokens."""
    n, L, Dt = text.shape
    if L > TEXT_PAD:
        raise ValueError("prompts of %d tokens exceed the %d-key text tile of the training path" % (L, TEXT_PAD))
    out = torch.zeros(n, TEXT_PAD, Dt, device=text.device, dtype=dtype)
    out[:, :L] = text.to(dtype)
    return out


# ---------------------------------------------------------------------------------------------- layer backward passes
def resnet_backward(r, x, skip, rowvec, dout, wsink=None, saved=None):
    """x [n, h, w, cx] (+ skip [n, h, w, cs]) as the forward saw them (the panorama branch: circularly padded by 2 columns, the
    layout the kept activations have too), dout fp32 [n, h, w, cout] -> (dx, dskip) fp32.
    wsink (trainable resnet, the ControlNet's): also the gradients of norm1 / conv1 / norm2 / conv2 / conv_shortcut into
    wsink(param, grad), and a third result: the gradient [n, cout] of this resnet's slice of the time-embedding projection."""
    tw = resnet_train(r, x.device)
    n, h, w, cx = x.shape
    hw, M = h * w, n * h * w
    cin = cx + (skip.shape[-1] if skip is not None else 0)
    if saved is not None:                             # kept by the training forward (engine.run_resnet(save=...))
        sc1, sh1, h1, sc2, sh2 = saved.sc1, saved.sh1, saved.h1.view(n, hw, r.cout), saved.sc2, saved.sh2
        y1 = ops.scale_shift_act(x, skip, n, hw, sc1, sh1, 1, out_dtype=r.dtype) if wsink is not None else None
    else:                                             # recompute: GN1 -> SiLU -> conv1 (+ bias + temb) -> GN2 statistics
        sc1, sh1 = ops.groupnorm_scale_shift(x, skip, n, hw, r.norm1.groups, r.norm1.eps, r.norm1.g, r.norm1.b)
        y1 = ops.scale_shift_act(x, skip, n, hw, sc1, sh1, 1, out_dtype=r.dtype)
        h1 = ops.conv_gemm(y1, r.w1, r.cout, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, bias=r.b1, rowvec=rowvec, out_dtype=F32)
        h1 = h1.view(n, hw, r.cout)
        sc2, sh2 = ops.groupnorm_scale_shift(h1, None, n, hw, r.norm2.groups, r.norm2.eps, r.norm2.g, r.norm2.b)
    # backward
    d, state = _normalise(dout.reshape(n, h, w, r.cout))
    dy2 = conv3_dgrad(d, tw.w2, r.cout, "s1", r.dtype)                                   # gradient of silu(gn2(h1))
    dh1, _ = ops.groupnorm_bwd(h1, None, n, hw, r.norm2.groups, r.norm2.eps, r.norm2.g, sc2, sh2, 1, dy2.view(n, hw, r.cout))
    dy1 = conv3_dgrad(dh1.view(n, h, w, r.cout), tw.w1, cin, "s1", r.dtype)              # gradient of silu(gn1(x | skip))
    if tw.ws is not None:
        dsc = stream_linear(d.view(M, r.cout), tw.ws, tw.ws3, cin, r.dtype)             # shortcut: d Ws  [M, cin]
    else:
        assert skip is None and cin == r.cout
        dsc = d
    dx, dskip = ops.groupnorm_bwd(x, skip, n, hw, r.norm1.groups, r.norm1.eps, r.norm1.g, sc1, sh1, 1,
                                  dy1.view(n, hw, cin), dres=dsc.view(n, hw, cin))
    dtemb = None
    if wsink is not None:
        src, put = r.src, _scaled(wsink, state)
        d16 = engine.to16(d, r.dtype)
        y2 = ops.scale_shift_act(h1, None, n, hw, sc2, sh2, 1, out_dtype=r.dtype).view(n, h, w, r.cout)
        put(src.conv2.weight, conv3_wgrad(d16, y2))
        put(src.conv2.bias, ops.colsum(d.view(M, r.cout)))
        g2, b2 = gn_param_grads(r.norm2, h1, None, n, hw, sc2, sh2, 1, dy2.view(n, hw, r.cout))
        put(src.norm2.weight, g2)
        put(src.norm2.bias, b2)
        dh16 = engine.to16(dh1.view(n, h, w, r.cout), r.dtype)
        put(src.conv1.weight, conv3_wgrad(dh16, y1.view(n, h, w, cin)))
        per_img = torch.stack([ops.colsum(dh1[i]) for i in range(n)], 0)                # [n, cout]: conv1's bias, temb rows
        put(src.conv1.bias, ops.colsum(per_img))
        dtemb = _unscale(per_img, state)
        g1, b1 = gn_param_grads(r.norm1, x, skip, n, hw, sc1, sh1, 1, dy1.view(n, hw, cin))
        put(src.norm1.weight, g1)
        put(src.norm1.bias, b1)
        if tw.ws is not None:
            ones, zeros = _unit_affine(cin, x.device)
            xs16 = ops.scale_shift_act(x, skip, n, hw, ones.expand(n, cin).contiguous(), zeros.expand(n, cin).contiguous(), 0,
                                       out_dtype=r.dtype)                                 # (x | skip) as one 16-bit operand
            put(src.conv_shortcut.weight, training.weight_grad(d16.view(M, r.cout), xs16.view(M, cin)))
            put(src.conv_shortcut.bias, ops.colsum(d.view(M, r.cout)))
    _unscale(dx, state)
    if dskip is not None:
        _unscale(dskip, state)
        dskip = dskip.view(n, h, w, -1)
    if wsink is not None:
        return dx.view(n, h, w, cx), dskip, dtemb
    return dx.view(n, h, w, cx), dskip


def _self_attention(tw, ln, n, hw, dh):
    """Recompute of attn1 on the normalised tokens ln [n*hw, C]: (qkv, qkv^T, V^T for the forward kernel, output, lse)."""
    Cc, H = tw.dim, tw.heads
    qkv = ops.linear(ln, tw.wqkv)                                                        # [T, 3C]
    qkv3 = qkv.view(n, hw, 3 * Cc)
    qkvt = ops.transpose_tokens(qkv3)                                                    # [n, 3C, hw]
    if hw % 32 == 0:
        vt = qkvt[:, 2 * Cc:]
        vt_ld, vt_bs = hw, 3 * Cc * hw
    else:                                           # the forward kernel reads V^T rows padded to 32 keys (4x4 level)
        vt = ops.linear_t(ln.view(n, hw, Cc), tw.wv)
        vt_ld, vt_bs = vt.shape[-1], vt.shape[1] * vt.shape[2]
    a = torch.empty(n, hw, Cc, device=ln.device, dtype=ln.dtype)
    lse = torch.empty(n, H, hw, device=ln.device, dtype=F32)
    ld = 3 * Cc
    ops.attention(qkv3[:, :, :Cc], qkv3[:, :, Cc:2 * Cc], vt, n, H, dh, hw, hw, q_ld=ld, k_ld=ld, vt_ld=vt_ld,
                  q_bs=hw * ld, k_bs=hw * ld, vt_bs=vt_bs, out=a, lse=lse)
    return qkv3, qkvt, a, lse


def transformer_recompute(t, x, text):
    """The block's forward up to the GEGLU input, keeping everything its backward reads.  Runs EITHER as the training forward
    itself (KEEP: TrainBranch.attention -> transformer_forward_keep; nothing is recomputed later) or inside the backward."""
    dev = x.device
    tw = transformer_train(t, dev)
    a1w, a2w = tw.attn1, tw.attn2
    n, h, w, Cc = x.shape
    hw, T = h * w, n * h * w
    H = a1w.heads
    dh = Cc // H
    L = text.shape[1]
    dt16 = t.dtype
    sc, sh = ops.groupnorm_scale_shift(x, None, n, hw, t.norm.groups, t.norm.eps, t.norm.g, t.norm.b)
    y = ops.scale_shift_act(x, None, n, hw, sc, sh, 0, out_dtype=dt16)
    if EXACT and t.w_in3 is not None:                 # as the mixed forward: proj_in carries the stream
        tok0 = engine.exact_gemm(engine.split_operand(x, None, sc, sh, 0, dtype=dt16), t.w_in3, Cc, w_in=T, bias=t.b_in, out_dtype=F32)
    else:
        tok0 = ops.linear(y.view(T, Cc), tw.w_in, bias=t.b_in, out_dtype=F32)
    ln1 = ops.layernorm(tok0, t.ln1.g, t.ln1.b, t.ln1.eps, out_dtype=dt16)
    qkv3, qkvt, a1, lse1 = _self_attention(a1w, ln1, n, hw, dh)
    tok1 = ops.linear(a1.view(T, Cc), a1w.wo, bias=a1w.bo, residual=tok0)
    ln2 = ops.layernorm(tok1, t.ln2.g, t.ln2.b, t.ln2.eps, out_dtype=dt16)
    q2 = ops.linear(ln2, a2w.wq).view(n, hw, Cc)
    q2t = ops.transpose_tokens(q2)
    textp = pad_text(text, dt16)                                                         # [n, 128, Dt]
    kv2 = ops.linear(textp.view(n * TEXT_PAD, -1), a2w.wkv).view(n, TEXT_PAD, 2 * Cc)    # (k | v)
    kv2t = ops.transpose_tokens(kv2)                                                     # [n, 2C, 128]
    tbias, tflags = text_bias(hw, L, dev)
    a2 = torch.empty(n, hw, Cc, device=dev, dtype=dt16)
    lse2 = torch.empty(n, H, hw, device=dev, dtype=F32)
    ops.attention(q2, kv2[:, :, :Cc], kv2t[:, Cc:], n, H, dh, hw, TEXT_PAD, q_ld=Cc, k_ld=2 * Cc, vt_ld=TEXT_PAD,
                  q_bs=hw * Cc, k_bs=TEXT_PAD * 2 * Cc, vt_bs=2 * Cc * TEXT_PAD, bias=tbias, flags=tflags, out=a2, lse=lse2)
    tok2 = ops.linear(a2.view(T, Cc), a2w.wo, bias=a2w.bo, residual=tok1)
    ln3 = ops.layernorm(tok2, t.ln3.g, t.ln3.b, t.ln3.eps, out_dtype=dt16)
    u = ops.linear(ln3, tw.w1, bias=tw.b1)                                               # [T, 8C] = (value | gate)
    return NS(sc=sc, sh=sh, y=y, tok0=tok0, ln1=ln1, qkv3=qkv3, qkvt=qkvt, a1=a1, lse1=lse1, tok1=tok1, ln2=ln2, q2=q2, q2t=q2t,
              textp=textp, kv2=kv2, kv2t=kv2t, tbias=tbias, tflags=tflags, a2=a2, lse2=lse2, tok2=tok2, ln3=ln3, u=u)


def transformer_forward_keep(t, x, text):
    """The training forward of a transformer block that KEEPS its activations: transformer_recompute, then GEGLU, FF2 and
    proj_out (+ the block's residual) exactly as engine.run_transformer ends.  -> (out [n, h, w, C] stream dtype, record).
    Against the inference forward the only differences are roundings: the GEGLU input is materialised in 16 bit (the fused
    epilogue gates the fp32 accumulators), the text keys are padded to 128 and masked."""
    n, h, w, Cc = x.shape
    T = n * h * w
    rec = transformer_recompute(t, x, text)
    tw = t.train
    g = ops.geglu(rec.u)
    if t.w_out3 is not None:          # mixed scheme: FF2's epilogue emits the [hi | lo] pair of the split-precision proj_out
        pair = ops.linear(g, t.w_ff2, bias=t.b_ff2, residual=rec.tok2, split_out=True)
        out = engine.exact_gemm(pair, t.w_out3, Cc, w_in=T, bias=t.b_out, residual=x.view(T, Cc), gn_stats=True)
    else:
        tok3 = ops.linear(g, t.w_ff2, bias=t.b_ff2, residual=rec.tok2)                   # (the kept token stream is fp32)
        out = ops.linear(engine.to16(tok3, t.dtype), t.w_out, bias=t.b_out, residual=x.view(T, Cc), gn_stats=True)
    return ops.carry(out.view(n, h, w, Cc), out), rec


def transformer_backward(t, x, text, dout, sink, wsink=None, rec=None):
    """x [n, h, w, C] as the forward saw it, text [n, L, Dt], dout fp32 [n, h, w, C] -> dx fp32; LoRA gradients into sink.
    wsink (trainable transformer, the ControlNet's): the gradient of every parameter of the block into wsink(param, grad).
    rec: the record the training forward kept (transformer_forward_keep); None: the block is recomputed here."""
    dev = x.device
    tw = transformer_train(t, dev)
    a1w, a2w = tw.attn1, tw.attn2
    n, h, w, Cc = x.shape
    hw, T = h * w, n * h * w
    H = a1w.heads
    dh = Cc // H
    dt16 = t.dtype
    if rec is None:
        rec = transformer_recompute(t, x, text)
    sc, sh, y, tok0, ln1, qkv3, qkvt, a1, lse1, tok1, ln2 = rec.sc, rec.sh, rec.y, rec.tok0, rec.ln1, rec.qkv3, rec.qkvt, rec.a1, rec.lse1, rec.tok1, rec.ln2
    q2, q2t, textp, kv2, kv2t, tbias, tflags, a2, lse2, tok2, ln3, u = rec.q2, rec.q2t, rec.textp, rec.kv2, rec.kv2t, rec.tbias, rec.tflags, rec.a2, rec.lse2, rec.tok2, rec.ln3, rec.u

    # ---- backward
    d, state = _normalise(dout.reshape(T, Cc))
    lsink = ScaledSink(sink, state)
    put = _scaled(wsink, state) if wsink is not None else None
    src = t.src
    blk = src.transformer_blocks[0]
    wg = training.weight_grad
    dtok3 = stream_linear(d, tw.w_out_t, tw.w_out_t3, Cc, dt16)                          # proj_out (its residual: d -> dx below)
    # feed-forward
    d3_16 = engine.to16(dtok3, dt16)
    dg = ops.linear(d3_16, tw.w2_t)
    du = ops.geglu_bwd(u, dg)
    dln3 = ops.linear(du, tw.w1_t, out_dtype=F32)
    dtok2, g_ln3, b_ln3 = ops.layernorm_bwd(tok2, t.ln3.g, dln3, t.ln3.eps, dres=dtok3)
    if put is not None:
        g = ops.geglu(u)                                                                 # [T, 4C]
        tok3 = ops.linear(g, tw.w2, bias=t.b_ff2, residual=tok2)                         # the input of proj_out
        put(src.proj_out.weight, wg(engine.to16(d, dt16), engine.to16(tok3, dt16)))
        put(src.proj_out.bias, ops.colsum(d))
        put(blk.ff.net[2].weight, wg(d3_16, g))
        put(blk.ff.net[2].bias, ops.colsum(dtok3))
        put(blk.ff.net[0].proj.weight, wg(du, ln3))
        put(blk.ff.net[0].proj.bias, ops.colsum(du))
        put(blk.norm3.weight, g_ln3)
        put(blk.norm3.bias, b_ln3)
    # text cross-attention
    d16 = engine.to16(dtok2, dt16)
    da2 = ops.linear(d16, a2w.wo_t).view(n, hw, Cc)
    if put is not None:
        put(blk.attn2.to_out[0].weight, wg(d16, a2.view(T, Cc)))
        put(blk.attn2.to_out[0].bias, ops.colsum(dtok2))
    if a2w.lora_out.live:
        a2w.lora_out.grads(a2.view(T, Cc), d16, lsink)
    delta2 = ops.attention_delta(a2, da2, n, H, dh, hw)
    dq2 = torch.empty(n, hw, Cc, device=dev, dtype=dt16)
    dkv2 = torch.empty(n, TEXT_PAD, 2 * Cc, device=dev, dtype=dt16)
    ops.attention_bwd(q2, kv2[:, :, :Cc], kv2[:, :, Cc:], da2, q2t, kv2t[:, :Cc], ops.transpose_tokens(da2), lse2, delta2,
                      dq2, dkv2[:, :, :Cc], dkv2[:, :, Cc:], n, H, dh, hw, TEXT_PAD,
                      q_ld=Cc, k_ld=2 * Cc, v_ld=2 * Cc, do_ld=Cc, dq_ld=Cc, dk_ld=2 * Cc, dv_ld=2 * Cc,
                      q_bs=hw * Cc, k_bs=TEXT_PAD * 2 * Cc, v_bs=TEXT_PAD * 2 * Cc, do_bs=hw * Cc, dq_bs=hw * Cc,
                      dk_bs=TEXT_PAD * 2 * Cc, dv_bs=TEXT_PAD * 2 * Cc, bias=tbias, flags=tflags)
    dln2 = ops.linear(dq2.view(T, Cc), a2w.wq_t, out_dtype=F32)
    if a2w.lora_q.live:
        a2w.lora_q.grads(ln2, dq2.view(T, Cc), lsink)
    if a2w.lora_kv.live:
        Tt = n * TEXT_PAD
        a2w.lora_kv.grads(textp.view(Tt, -1), dkv2.view(Tt, 2 * Cc), lsink)
    dtok1, g_ln2, b_ln2 = ops.layernorm_bwd(tok1, t.ln2.g, dln2, t.ln2.eps, dres=dtok2)
    if put is not None:
        put(blk.attn2.to_q.weight, wg(dq2.view(T, Cc), ln2))
        Tt = n * TEXT_PAD
        dwkv = wg(dkv2.view(Tt, 2 * Cc), textp.view(Tt, -1))                             # [2C, Dt] = (k | v) rows
        put(blk.attn2.to_k.weight, dwkv[:Cc])
        put(blk.attn2.to_v.weight, dwkv[Cc:])
        put(blk.norm2.weight, g_ln2)
        put(blk.norm2.bias, b_ln2)
    # self-attention
    d16 = engine.to16(dtok1, dt16)
    da1 = ops.linear(d16, a1w.wo_t).view(n, hw, Cc)
    if put is not None:
        put(blk.attn1.to_out[0].weight, wg(d16, a1.view(T, Cc)))
        put(blk.attn1.to_out[0].bias, ops.colsum(dtok1))
    if a1w.lora_out.live:
        a1w.lora_out.grads(a1.view(T, Cc), d16, lsink)
    delta1 = ops.attention_delta(a1, da1, n, H, dh, hw)
    dqkv = torch.empty(n, hw, 3 * Cc, device=dev, dtype=dt16)
    ld = 3 * Cc
    ops.attention_bwd(qkv3[:, :, :Cc], qkv3[:, :, Cc:2 * Cc], qkv3[:, :, 2 * Cc:], da1, qkvt[:, :Cc], qkvt[:, Cc:2 * Cc],
                      ops.transpose_tokens(da1), lse1, delta1, dqkv[:, :, :Cc], dqkv[:, :, Cc:2 * Cc], dqkv[:, :, 2 * Cc:],
                      n, H, dh, hw, hw, q_ld=ld, k_ld=ld, v_ld=ld, do_ld=Cc, dq_ld=ld, dk_ld=ld, dv_ld=ld,
                      q_bs=hw * ld, k_bs=hw * ld, v_bs=hw * ld, do_bs=hw * Cc, dq_bs=hw * ld, dk_bs=hw * ld, dv_bs=hw * ld)
    dln1 = ops.linear(dqkv.view(T, 3 * Cc), a1w.wqkv_t, out_dtype=F32)
    if a1w.lora_qkv.live:
        a1w.lora_qkv.grads(ln1, dqkv.view(T, 3 * Cc), lsink)
    dtok0, g_ln1, b_ln1 = ops.layernorm_bwd(tok0, t.ln1.g, dln1, t.ln1.eps, dres=dtok1)
    # proj_in and the GroupNorm in front of it; the block's own residual (out = proj_out(..) + x)
    dy = stream_linear(dtok0, tw.w_in_t, tw.w_in_t3, Cc, dt16)
    if put is not None:
        dwqkv = wg(dqkv.view(T, 3 * Cc), ln1)                                            # [3C, C] = (q | k | v) rows
        put(blk.attn1.to_q.weight, dwqkv[:Cc])
        put(blk.attn1.to_k.weight, dwqkv[Cc:2 * Cc])
        put(blk.attn1.to_v.weight, dwqkv[2 * Cc:])
        put(blk.norm1.weight, g_ln1)
        put(blk.norm1.bias, b_ln1)
        put(src.proj_in.weight, wg(engine.to16(dtok0, dt16), y.view(T, Cc)))
        put(src.proj_in.bias, ops.colsum(dtok0))
        g_n, b_n = gn_param_grads(t.norm, x, None, n, hw, sc, sh, 0, dy.view(n, hw, Cc))
        put(src.norm.weight, g_n)
        put(src.norm.bias, b_n)
    dx, _ = ops.groupnorm_bwd(x, None, n, hw, t.norm.groups, t.norm.eps, t.norm.g, sc, sh, 0, dy.view(n, hw, Cc), dres=d.view(n, hw, Cc))
    return _unscale(dx, state).view(n, h, w, Cc)


class ScaledSink:
    """The parameter sink of one block's LoRA groups together with the block's gradient-normalisation state: the groups'
    reductions (LoRAGroup.grads -> pf_weighted_colsum) scale their results back on the device by state[2] = 2^e."""

    def __init__(self, sink, state):
        self.sink, self.state = sink, state


def downsample_backward(d, dout, pano_pad, dev_dtype, wsink=None, x=None):
    """dout fp32 [n, ho, wo, C] -> dx of the block input; panorama: pad 2 / conv s2 / crop 1 (MVGenModel.py:138-144).
    wsink + x (the conv's input [n, h, w, cin]): also the gradients of the trainable conv's weight and bias."""
    tw = getattr(d, "train", None)
    cin = d.src.weight.shape[1]
    if tw is None:
        wflip = d.src.weight.detach().float().flip(2, 3).permute(1, 2, 3, 0).reshape(cin, -1)      # [cin, 9 * cout]
        tw = d.train = NS(w=wflip.to(device=dout.device, dtype=dev_dtype).contiguous(), w3=None)
        if EXACT:
            tw.w3 = engine._split_weight(wflip, 9, dout.device, dev_dtype)
    g, state = _normalise(dout)
    if pano_pad:
        g = ops.crop_width_bwd(g, 1)
    if wsink is not None:
        assert not pano_pad and x is not None
        put = _scaled(wsink, state)
        put(d.src.weight, conv3_wgrad(engine.to16(g, dev_dtype), engine.to16(x, dev_dtype), stride=2))
        put(d.src.bias, ops.colsum(g.view(-1, g.shape[-1])))
    if tw.w3 is not None:                             # the stride-2 conv maps the stream onto itself: split precision
        n, ho, wo, cout = g.shape
        z = ops.zero_insert2(engine.split_operand(g, dtype=dev_dtype).view(n, ho, wo, 2 * cout))
        dx = engine.exact_gemm(z, tw.w3, cin, n_img=n, h_in=2 * ho, w_in=2 * wo, ksize=3, pad=1, out_dtype=F32).view(n, 2 * ho, 2 * wo, cin)
    else:
        dx = conv3_dgrad(g, tw.w, cin, "s2", dev_dtype)
    if pano_pad:
        dx = ops.pad_width_bwd(dx, 2)
    return _unscale(dx, state)


def upsample_backward(up, dout, pano_pad, dev_dtype):
    """panorama: pad 1 / nearest x2 + conv / crop 2 (MVGenModel.py:272-277)."""
    tw = getattr(up, "train", None)
    if tw is None:
        tw = up.train = NS(w=flip_conv3_weight(up.src.weight, dev_dtype))
    g, state = _normalise(dout)
    if pano_pad:
        g = ops.crop_width_bwd(g, 2)
    dx = conv3_dgrad(g, tw.w.to(dout.device), up.src.weight.shape[1], "up", dev_dtype)
    if pano_pad:
        dx = ops.pad_width_bwd(dx, 1)
    return _unscale(dx, state)


def head_backward(u, hin, d_eps, pano_pad):
    """hin [n, h, w, C] the head's input, d_eps fp32 NCHW gradient of the predicted noise -> fp32 [n, h, w, C]."""
    n, h, w, Cc = hin.shape
    tw = getattr(u, "train_head", None)
    if tw is None:
        tw = u.train_head = NS(w=conv_out_dgrad_weight(u.src_conv_out.weight).to(hin.device))
    g, state = _normalise(d_eps.float().contiguous())
    sc, sh = ops.groupnorm_scale_shift(hin, None, n, h * w, u.norm_out.groups, u.norm_out.eps, u.norm_out.g, u.norm_out.b)
    dy = conv_out_dgrad(g, tw.w, Cc, bool(pano_pad))
    dx, _ = ops.groupnorm_bwd(hin, None, n, h * w, u.norm_out.groups, u.norm_out.eps, u.norm_out.g, sc, sh, 1, dy.view(n, h * w, Cc))
    return _unscale(dx, state).view(n, h, w, Cc)


# ---------------------------------------------------------------------------------------------- the tape
class TrainBranch(engine.Branch):
    """engine.Branch that notes which packed layer ran on which tensors (nothing else changes in the forward)."""

    def __init__(self, tape, *args, **kw):
        super().__init__(*args, **kw)
        self.tape = tape

    def resnet(self, r, skip=False):
        x, s = self.h, (self.skips[-1] if skip else None)
        saved = NS() if KEEP else None
        super().resnet(r, skip, save=saved)
        self.tape.append(("resnet", self, r, x, s, saved))

    def attention(self, t):
        x = self.h
        rec = None
        if KEEP:
            self.h, rec = transformer_forward_keep(t, x, self.text)
        else:
            super().attention(t)
        self.tape.append(("attention", self, t, x, rec))

    def push(self):
        super().push()
        self.tape.append(("push", self))

    def downsample(self, d):
        x = self.h
        super().downsample(d)
        self.tape.append(("down", self, d, x))

    def upsample(self, up):
        super().upsample(up)
        self.tape.append(("up", self, up))

    def head(self):
        x = self.h
        out = super().head()
        self.tape.append(("head", self, x))
        return out


class ParamGrads:
    """Gradient sink: parameter tensor -> accumulated fp32 gradient (device tensors; torch adds are bookkeeping on tiny
    LoRA / EPA matrices, the heavy reductions ran in the GEMM kernel)."""

    def __init__(self):
        self.grads = {}

    def __call__(self, param, grad, scale=1.0):
        g = grad.reshape(param.shape).to(F32)
        if scale != 1.0:
            g = g * scale
        key = id(param)
        if key in self.grads:
            self.grads[key] = (param, ops.add(self.grads[key][1].contiguous(), g.contiguous()))
        else:
            self.grads[key] = (param, g)

    def get(self, param):
        hit = self.grads.get(id(param))
        return None if hit is None else hit[1].to(device=param.device, dtype=param.dtype)      # (modules may live on the host)


def _trains(refs):
    return any(r is not None and (r.up.requires_grad or r.down.requires_grad) for r in refs)


def _first_trainable_entry(tape):
    """Index of the earliest tape entry whose backward feeds a parameter that requires a gradient: an EPA block, a ControlNet
    marker, a transformer with a trainable LoRA pair.  None: nothing on the tape trains."""
    for i, entry in enumerate(tape):
        kind = entry[0]
        if kind in ("cn_skips", "cn_mid"):
            return i
        if kind == "fuse" and any(p_.requires_grad for p_ in training.train_params(entry[3])):
            return i
        if kind == "attention":
            tw = transformer_train(entry[2], entry[3].device)
            groups = (tw.attn1.lora_qkv, tw.attn1.lora_out, tw.attn2.lora_q, tw.attn2.lora_kv, tw.attn2.lora_out)
            if any(_trains(g.refs) for g in groups):
                return i
    return None


def backward(tape, d_eps, sink, dh=None, dskips=None, wsink=None, dtemb=None, side=None):
    """Walk the tape backwards.  d_eps: {branch: fp32 NCHW gradient of that branch's predicted noise}.
    dh / dskips: gradients to start from (a ControlNet's tape starts at its mid output and its 12 skip tensors);
    wsink: the taped network's own weights train -- parameter gradients into wsink, each resnet's time-embedding gradient
    appended to dtemb as (offset, [n, cout]).
    side: the stream the forward ran the panorama branch on (entries of a branch with `on_side` are walked there, the EPA blocks
    join the two streams as in the forward); tensors handed from one stream to the other stay referenced until the walk ends
    (the caching allocator recycles per stream)."""
    import contextlib
    dh = {} if dh is None else dh
    dskips = {} if dskips is None else dskips
    main = torch.cuda.current_stream() if side is not None else None
    crossing = []
    if side is not None:
        side.wait_stream(main)

    def stream_of(br):
        return torch.cuda.stream(side) if (side is not None and getattr(br, "on_side", False)) else contextlib.nullcontext()
    # Nothing before the EARLIEST entry that has a trainable leaf behind it needs a gradient (autograd prunes the same way):
    # with frozen LoRA matrices (layout-conditioned runs) the first encoder level of both branches is skipped.
    first = 0 if wsink is not None else _first_trainable_entry(tape)
    if first is None:
        return dh, dskips
    for index in range(len(tape) - 1, first - 1, -1):
        entry = tape[index]
        kind, br = entry[0], entry[1]
        if kind == "cn_mid":                          # h += mid residual (MVGenModel.py:200-203): its gradient is dh as it stands
            entry[2].d_mid = dh[br]
            continue
        if kind == "cn_skips":                        # skips += residuals (:154-170): every skip's gradient is known by now
            with stream_of(br):
                controlnet_backward(entry[2], list(dskips[br]), entry[2].d_mid, sink)
            continue
        if kind == "fuse":
            _, pers, pano, block, xp, xe, groups, m, rec = entry
            if side is not None:
                main.wait_stream(side)                # join: the panorama branch's gradient is complete
                crossing.append(dh[pano])
            dp, de, grads = block.backward_nhwc(xp, xe, groups, m, dh[pers], dh[pano], rec)
            dh[pers], dh[pano] = dp, de
            for p_, g_ in zip(training.train_params(block), grads):
                sink(p_, g_)
            if side is not None:
                crossing.append(de)
                side.wait_stream(main)                # fork
            continue
        with stream_of(br):
            _walk_entry(entry, kind, br, dh, dskips, d_eps, sink, wsink, dtemb)
    if side is not None:
        main.wait_stream(side)
    del crossing
    return dh, dskips


def _walk_entry(entry, kind, br, dh, dskips, d_eps, sink, wsink, dtemb):
    """One UNet-layer entry of the tape (on the stream the caller selected)."""
    pad = br.pad
    if kind == "head":
        dh[br] = head_backward(br.u, entry[2], d_eps[br], pad)
    elif kind == "up":
        dh[br] = upsample_backward(entry[2], dh[br], pad, br.u.dtype)
    elif kind == "down":
        dh[br] = downsample_backward(entry[2], dh[br], pad, br.u.dtype, wsink, entry[3] if wsink is not None else None)
    elif kind == "push":
        g = dskips[br].pop()
        dh[br] = ops.add(dh[br], g) if dh.get(br) is not None else g
    elif kind == "attention":
        dh[br] = transformer_backward(entry[2], entry[3], br.text, dh[br], sink, wsink, entry[4])
    elif kind == "resnet":
        _, _, r, x, s, saved = entry
        rowvec = br.temb[:, r.temb_off:]
        d = dh[br]
        if pad:                                   # pad 2 / resnet / crop 2 (MVGenModel.py:110-115)
            d = ops.crop_width_bwd(d, 2)
            x, s = ops.pad_width(x, 2), (ops.pad_width(s, 2) if s is not None else None)
        if wsink is not None:
            dx, ds, dt = resnet_backward(r, x, s, rowvec, d, wsink, saved)
            dtemb.append((r.temb_off, dt))
        else:
            dx, ds = resnet_backward(r, x, s, rowvec, d, saved=saved)
        if pad:
            dx, ds = ops.pad_width_bwd(dx, 2), (ops.pad_width_bwd(ds, 2) if ds is not None else None)
        dh[br] = dx
        if s is not None:
            dskips.setdefault(br, []).append(ds)


# ---------------------------------------------------------------------------------------------- the trainable ControlNet
def cond_embedding_train(c, cond, rec):
    """engine.run_cond_embedding (diffusers ControlNetConditioningEmbedding.forward) keeping every layer's 16-bit input
    (rec.xs) and pre-activation (rec.zs) for the backward; channel counts zero-padded to multiples of 64 as there."""
    n, _, H, W = cond.shape
    dev = cond.device
    z = ops.conv_in(cond.float(), c.ce_w0, c.ce_b0, c.ce_c0, c.dtype, wrap=False)
    rec.zs, rec.xs = [z], []
    x = ops.silu(z)
    h, w = H, W
    for i, L in enumerate(c.ce_layers):
        ho, wo = (h - 1) // L.stride + 1, (w - 1) // L.stride + 1
        out = torch.zeros(n, ho, wo, engine._pad64(L.cout), device=dev, dtype=c.dtype)
        ops.conv_gemm(x, L.w, L.cout, n_img=n, h_in=h, w_in=w, ksize=3, stride=L.stride, pad=1, bias=L.b,
                      c0=L.cin_pad, out=out.view(-1, out.shape[-1]))
        rec.xs.append(x)
        if i + 1 < len(c.ce_layers):
            rec.zs.append(out)
            x = ops.silu(out)
        else:
            x = out
        h, w = ho, wo
    return x


def controlnet_forward(c, latent, timestep, text, cond):
    """The ControlNet's forward of a training step: engine.run_controlnet on a taping branch.  -> (residuals, record)."""
    rec = NS(c=c, tape=[], latent=latent, timestep=timestep, cond=cond)
    res = engine.run_controlnet(c, latent, timestep, text, cond, rec=rec,
                                make_branch=lambda *a, **k: TrainBranch(rec.tape, *a, **k),
                                embed=lambda c_, cond_: cond_embedding_train(c_, cond_, rec))
    return res, rec


def _nhwc8(x_nchw, dtype):
    """NCHW fp32 with < 8 channels (the 4-channel latent, the 3-channel layout image) -> NHWC 16-bit, channels zero-padded
    to 8: the operand of pf_im2col3 for the weight gradient of a boundary convolution."""
    n, ch, h, w = x_nchw.shape
    out = torch.zeros(n, h, w, 8, device=x_nchw.device, dtype=dtype)
    out[..., :ch] = x_nchw.permute(0, 2, 3, 1).to(dtype)
    return out


def _padded_flip(conv, cin_pad, cout_pad, dev, dtype):
    """Data-gradient operand (flip_conv3_weight) of a conv whose channel counts are zero-padded to the buffers' strides."""
    co, ci = conv.weight.shape[:2]
    w = torch.zeros(cout_pad, cin_pad, 3, 3, device=dev, dtype=F32)
    w[:co, :ci] = conv.weight.detach().to(device=dev, dtype=F32)
    return flip_conv3_weight(w, dtype)


def cond_embedding_backward(c, rec, d_out, sink):
    """d_out fp32 [n, h, w, c0]: gradient of the conditioning embedding's output -> gradients of its 8 convolutions."""
    ce = c.src.controlnet_cond_embedding
    convs = [*ce.blocks, ce.conv_out]
    d = d_out
    for i in reversed(range(len(c.ce_layers))):
        L, conv, x = c.ce_layers[i], convs[i], rec.xs[i]
        cin_pad, cout_pad = x.shape[-1], d.shape[-1]
        g, state = _normalise(d)
        put = _scaled(sink, state)
        put(conv.weight, conv3_wgrad(engine.to16(g, c.dtype), x, stride=L.stride)[:L.cout, :conv.weight.shape[1]])
        put(conv.bias, ops.colsum(g.view(-1, cout_pad))[:L.cout])
        wflip = getattr(L, "wflip", None)
        if wflip is None:
            wflip = L.wflip = _padded_flip(conv, cin_pad, cout_pad, x.device, c.dtype)
        dx = conv3_dgrad(g, wflip, cin_pad, "s2" if L.stride == 2 else "s1", c.dtype)
        d = _unscale(ops.silu_bwd(rec.zs[i], dx), state)
    g, state = _normalise(d)                              # the boundary conv 3 -> 16 (fp32 kernel in the forward)
    put = _scaled(sink, state)
    co, ci = ce.conv_in.weight.shape[:2]
    put(ce.conv_in.weight, conv3_wgrad(engine.to16(g, c.dtype), _nhwc8(rec.cond.float(), c.dtype))[:co, :ci])
    put(ce.conv_in.bias, ops.colsum(g.view(-1, g.shape[-1]))[:co])


def time_embedding_backward(c, timestep, dtemb, sink):
    """dtemb fp32 [n, temb_total]: gradient of the concatenated time_emb_proj outputs (engine.Branch.temb) -> gradients of
    every resnet's time_emb_proj and of TimestepEmbedding's two linears (Linear - SiLU - Linear, then SiLU - proj)."""
    dt = c.dtype
    te = c.src.time_embedding
    feats = ops.timestep_features(timestep, c.t_dim, dt)
    e1 = ops.linear(feats, c.w_t1, bias=c.b_t1)
    s1 = ops.silu(e1)
    e2 = ops.linear(s1, c.w_t2, bias=c.b_t2)
    s2 = ops.silu(e2)
    g, state = _normalise(dtemb)
    put = _scaled(sink, state)
    g16 = engine.to16(g, dt)
    dw, db = training.weight_grad(g16, s2), ops.colsum(g)
    for blk in [*c.down, c.mid]:
        for r in blk.resnets:
            put(r.src.time_emb_proj.weight, dw[r.temb_off:r.temb_off + r.cout])
            put(r.src.time_emb_proj.bias, db[r.temb_off:r.temb_off + r.cout])
    ds2 = _unscale(ops.linear(g16, c.w_temb.t().contiguous(), out_dtype=F32), state)
    g, state = _normalise(ops.silu_bwd(e2, ds2))
    put = _scaled(sink, state)
    g16 = engine.to16(g, dt)
    put(te.linear_2.weight, training.weight_grad(g16, s1))
    put(te.linear_2.bias, ops.colsum(g))
    ds1 = _unscale(ops.linear(g16, c.w_t2.t().contiguous(), out_dtype=F32), state)
    g, state = _normalise(ops.silu_bwd(e1, ds1))
    put = _scaled(sink, state)
    put(te.linear_1.weight, training.weight_grad(engine.to16(g, dt), feats))
    put(te.linear_1.bias, ops.colsum(g))


def controlnet_backward(rec, d_skips, d_mid, sink):
    """Gradients of EVERY ControlNet parameter (the reference's trainable set under layout conditions,
    PanoGenerator.py:153-157) from the gradients of its 12 skip residuals and its mid residual (fp32 NHWC)."""
    c, br = rec.c, rec.br
    cn = c.src
    dt = c.dtype

    def zero_conv_backward(z, mod, x, d):
        n, h, w, Cc = x.shape
        M = n * h * w
        g, state = _normalise(d.reshape(M, z.c))
        put = _scaled(sink, state)
        g16 = engine.to16(g, dt)
        put(mod.weight, training.weight_grad(g16, engine.to16(x, dt).view(M, Cc)))
        put(mod.bias, ops.colsum(g))
        wt = getattr(z, "wt", None)
        if wt is None:
            wt = z.wt = z.w.t().contiguous()
        return _unscale(ops.linear(g16, wt, out_dtype=F32), state).view(n, h, w, Cc)

    dh = {br: zero_conv_backward(c.zero_mid, cn.controlnet_mid_block, rec.h_mid, d_mid)}
    dsk = {br: [zero_conv_backward(z, mod, x, d) for z, mod, x, d in zip(c.zero_down, cn.controlnet_down_blocks, rec.skips, d_skips)]}
    dtemb = []
    dh, dsk = backward(rec.tape, {}, sink, dh=dh, dskips=dsk, wsink=sink, dtemb=dtemb)
    assert len(dsk[br]) == 1                                  # the first skip: conv_in(latent) + conditioning embedding
    d0 = ops.add(dh[br], dsk[br][0])
    g, state = _normalise(d0)
    put = _scaled(sink, state)
    co, ci = cn.conv_in.weight.shape[:2]
    put(cn.conv_in.weight, conv3_wgrad(engine.to16(g, dt), _nhwc8(rec.latent.float(), dt))[:, :ci])
    put(cn.conv_in.bias, ops.colsum(g.view(-1, co)))
    cond_embedding_backward(c, rec, d0, sink)
    dtemb.sort(key=lambda e: e[0])
    time_embedding_backward(c, rec.timestep, torch.cat([t for _, t in dtemb], 1).contiguous(), sink)


class DenoiserFunction(torch.autograd.Function):
    """``MultiViewBaseModel.forward`` under autograd: the inference kernels forward (with a tape of layer inputs), the
    tape walked backwards in ``backward``.  Gradients go to the EPA parameters and the LoRA matrices; the inputs
    (latents, prompts) take none -- the reference's training step does not ask for them."""

    @staticmethod
    def forward(ctx, model, args, *params):
        tape = []
        model.refold_lora()
        sample, pano_sample, pers, pano, side = model._forward(*args, tape=tape)
        ctx.tape, ctx.pers, ctx.pano, ctx.params, ctx.side = tape, pers, pano, params, side
        ctx.has_sample = sample is not None
        ctx.sample_shape = None if sample is None else tuple(sample.shape)
        ctx.pano_shape = tuple(pano_sample.shape)
        if sample is None:
            return pano_sample
        return sample, pano_sample

    @staticmethod
    def backward(ctx, *douts):
        d_sample, d_pano = (douts if ctx.has_sample else (None, douts[0]))
        d_eps = {}
        with torch.no_grad():
            if ctx.pers is not None:
                if d_sample is None:
                    d_sample = torch.zeros(ctx.sample_shape, device=d_pano.device, dtype=F32)
                d_eps[ctx.pers] = d_sample.flatten(0, 1).float().contiguous()
            if d_pano is None:                  # (a loss on the views only)
                d_pano = torch.zeros(ctx.pano_shape, device=d_sample.device, dtype=F32)
            d_eps[ctx.pano] = d_pano.flatten(0, 1).float().contiguous()
            sink = ParamGrads()
            backward(ctx.tape, d_eps, sink, side=ctx.side)
        # the tape (and with it every kept activation) must not outlive this call: the branches reference the same list, and the
        # graph node stays alive for as long as the caller holds the step's loss -- typically into the next step's forward
        ctx.tape.clear()
        ctx.tape = ctx.pers = ctx.pano = None
        return (None, None, *[sink.get(p_) for p_ in ctx.params])
