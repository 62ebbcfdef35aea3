"""Training path of the EPA block: backward of ``WarpAttn`` on the HIP kernels.

Reference: ``models/pano/modules.py:15-59`` under autograd, with the transformer block wrapped in the
reference's ``CheckpointFunction`` (``models/modules/transformer.py:77-127,137-161``): nothing but the block inputs
is kept from the forward, the block is recomputed in backward.  The same here: ``WarpAttn.forward`` runs the
inference path (engine.run_epa) and keeps ``pers_x`` / ``equi_x``; ``epa_backward`` recomputes the activations it
needs and walks the block backwards.

Both directions of the block (panorama queries over view keys, view queries over panorama keys) share their weights,
so every per-token operation runs ONCE on the concatenated token array [equi tokens | view tokens]; only the two
attentions are separate launches.

    forward (recomputed)                         backward
    n   = LN1(x + pe)                            dx  = dy + LN1'(dn)            dgamma1, dbeta1
    qkv = n Wqkv^T                               dn  = dqkv Wqkv                dWqkv = dqkv^T n
    a   = attention(q, k', v', bias)             dq, dk', dv' = attention'(da)  (k', v': the other token set)
    y   = a Wo^T + bo + x                        da  = dy Wo                    dWo = dy^T a, dbo = colsum(dy)
    z   = LN2(y)                                 dy  = dout + LN2'(dz)          dgamma2, dbeta2
    u   = z W1^T + b1                            dz  = du W1                    dW1 = du^T z, db1 = colsum(du)
    g   = u_a * gelu(u_gate)                     du  = GEGLU'(u, dg)
    out = g W2^T + b2 + y                        dg  = dout W2                  dW2 = dout^T g, db2 = colsum(dout)

MFMA operands are 16-bit (the block's compute dtype), every accumulation, the residual-stream gradients (dout, dy,
dx), the LayerNorm backward and all parameter gradients are fp32.  The incoming gradients are normalised on the
device by a power of two (max |dout| in [1, 2)) before they become 16-bit operands and the results are scaled back at
the end: an MSE loss over ~1e5 latent elements hands down gradients of ~1e-5, below fp16's normal range.
"""
import torch

from . import engine, ops

NS = engine.NS


def train_params(block):
    """The block's trainable tensors in the order ``epa_backward`` returns their gradients."""
    tr = block.transformer
    a, ff = tr.attn1, tr.ff.net
    return [tr.norm1.weight, tr.norm1.bias, a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_out.weight, a.to_out.bias,
            ff[0].proj.weight, ff[0].proj.bias, ff[2].weight, ff[2].bias, tr.norm2.weight, tr.norm2.bias]


def pack_epa_train(block, dev, dtype):
    """16-bit copies of the weights in both orientations (forward GEMMs read W [N, K], the data-gradient GEMMs W^T)."""
    tr = block.transformer
    a, ff = tr.attn1, tr.ff.net
    w = lambda t: t.detach().to(device=dev, dtype=torch.float32)
    f16 = lambda t: t.to(dtype).contiguous()
    e = NS()
    e.dtype, e.dim = dtype, tr.norm1.weight.shape[0]
    e.heads = e.dim // 32
    wqkv = torch.cat([w(a.to_q.weight), w(a.to_k.weight), w(a.to_v.weight)], 0)
    e.wqkv, e.wqkv_t = f16(wqkv), f16(wqkv.t())
    e.wv = f16(w(a.to_v.weight))
    e.wo, e.wo_t, e.bo = f16(w(a.to_out.weight)), f16(w(a.to_out.weight).t()), w(a.to_out.bias).contiguous()
    e.w1, e.w1_t, e.b1 = f16(w(ff[0].proj.weight)), f16(w(ff[0].proj.weight).t()), w(ff[0].proj.bias).contiguous()
    e.w2, e.w2_t, e.b2 = f16(w(ff[2].weight)), f16(w(ff[2].weight).t()), w(ff[2].bias).contiguous()
    e.ln1 = NS(g=w(tr.norm1.weight).contiguous(), b=w(tr.norm1.bias).contiguous(), eps=tr.norm1.eps)
    e.ln2 = NS(g=w(tr.norm2.weight).contiguous(), b=w(tr.norm2.bias).contiguous(), eps=tr.norm2.eps)
    return e


def weight_grad(dy16, x16):
    """dW [N, K] = dy^T x over the token rows: dy16 [T, N], x16 [T, K] 16-bit -> fp32.  Both operands are put
    token-contiguous (the reduction axis of the MFMA GEMM) by a transpose pass."""
    T, N = dy16.shape
    K = x16.shape[1]
    if T % 64:                        # the GEMM's reduction length is a multiple of 64: zero rows change nothing
        Tp = (T + 63) // 64 * 64
        pad = lambda t: torch.cat([t, torch.zeros(Tp - T, t.shape[1], device=t.device, dtype=t.dtype)], 0)
        dy16, x16, T = pad(dy16), pad(x16), Tp
    dyt = ops.transpose_tokens(dy16.view(1, T, N)).view(N, T)
    xt = ops.transpose_tokens(x16.view(1, T, K)).view(K, T)
    return ops.conv_gemm(dyt, xt, K, w_in=N, out_dtype=torch.float32)


def _attention_calls(b, shared, tables):
    """(batch slice, batch count, table) per attention launch: one launch when every sample has the same cameras."""
    if shared:
        return [(0, b, tables[0])]
    return [(i, 1, t) for i, t in enumerate(tables)]


def epa_recompute(e, tables, xe, xp, b, m):
    """Forward of the block up to the GEGLU output, keeping what backward reads.
    xe [b*E, C], xp [b*m*P, C] token rows (fp32 or 16-bit); tables: one EPATables entry, or one per sample."""
    Cc, H = e.dim, e.heads
    E, mP = xe.shape[0] // b, xp.shape[0] // b
    Te = b * E
    s = NS(E=E, mP=mP, Te=Te, T=Te + b * mP, shared=len(tables) == 1)
    x = torch.cat([xe, xp], 0)
    ln = torch.empty(s.T, Cc, device=x.device, dtype=e.dtype)
    for i in range(b):
        t = tables[0] if s.shared else tables[i]
        ops.layernorm(x[i * E:(i + 1) * E], e.ln1.g, e.ln1.b, e.ln1.eps, pe=t.pe_e, out=ln[i * E:(i + 1) * E])
        ops.layernorm(x[Te + i * mP:Te + (i + 1) * mP], e.ln1.g, e.ln1.b, e.ln1.eps, pe=t.pe_p,
                      out=ln[Te + i * mP:Te + (i + 1) * mP])
    qkv = ops.linear(ln, e.wqkv)                                         # [T, 3C] = (q | k | v)
    qkv_e, qkv_p = qkv[:Te].view(b, E, 3 * Cc), qkv[Te:].view(b, mP, 3 * Cc)
    qkvt_e, qkvt_p = ops.transpose_tokens(qkv_e), ops.transpose_tokens(qkv_p)      # [b, 3C, tokens]
    a = torch.empty(s.T, Cc, device=x.device, dtype=e.dtype)
    a_e, a_p = a[:Te].view(b, E, Cc), a[Te:].view(b, mP, Cc)
    lse_e = torch.empty(b, H, E, device=x.device, dtype=torch.float32)
    lse_p = torch.empty(b, H, mP, device=x.device, dtype=torch.float32)
    ld = 3 * Cc

    def values_t(qkvt_seg, ln_seg, n_tok):
        """V^T [b, C, ld] for the forward kernel: a row slice of the (q | k | v) transpose, or -- token counts that are
        not multiples of 32 (tiny test geometries) -- the transposed projection with rows padded to 32 keys."""
        if n_tok % 32 == 0:
            return qkvt_seg[:, 2 * Cc:], n_tok, ld * n_tok
        vt = ops.linear_t(ln_seg.view(b, n_tok, Cc), e.wv)
        return vt, vt.shape[-1], vt.shape[1] * vt.shape[2]
    vt_p, vt_p_ld, vt_p_bs = values_t(qkvt_p, ln[Te:], mP)
    vt_e, vt_e_ld, vt_e_bs = values_t(qkvt_e, ln[:Te], E)
    for i, n, t in _attention_calls(b, s.shared, tables):
        sl = slice(i, i + n)
        # panorama pixels query the views (modules.py:43-48) ...
        ops.attention(qkv_e[sl, :, :Cc], qkv_p[sl, :, Cc:2 * Cc], vt_p[sl], n, H, 32, E, mP,
                      q_ld=ld, k_ld=ld, vt_ld=vt_p_ld, q_bs=E * ld, k_bs=mP * ld, vt_bs=vt_p_bs,
                      bias=t.bias_e, flags=t.flags_e, out=a_e[sl], lse=lse_e[sl])
        # ... and the views query the panorama (modules.py:50-55)
        ops.attention(qkv_p[sl, :, :Cc], qkv_e[sl, :, Cc:2 * Cc], vt_e[sl], n, H, 32, mP, E,
                      q_ld=ld, k_ld=ld, vt_ld=vt_e_ld, q_bs=mP * ld, k_bs=E * ld, vt_bs=vt_e_bs,
                      bias=t.bias_p, flags=t.flags_p, out=a_p[sl], lse=lse_p[sl])
    y = ops.linear(a, e.wo, bias=e.bo, residual=x)                       # stream dtype of x
    z = ops.layernorm(y, e.ln2.g, e.ln2.b, e.ln2.eps, out_dtype=e.dtype)
    u = ops.linear(z, e.w1, bias=e.b1)                                   # [T, 8C] = (value | gate)
    g = ops.geglu(u)
    s.x, s.ln, s.qkv, s.qkvt_e, s.qkvt_p, s.a, s.lse_e, s.lse_p, s.y, s.z, s.u, s.g = \
        x, ln, qkv, qkvt_e, qkvt_p, a, lse_e, lse_p, y, z, u, g
    return s


def epa_backward(e, tables, s, d_e, d_p, b, m):
    """Gradients of the block.  d_e [b*E, C], d_p [b*m*P, C]: fp32 gradients of the two outputs (token rows);
    s: epa_recompute's record.  Returns (dx_e, dx_p, [parameter gradients in train_params order]), all fp32."""
    Cc, H = e.dim, e.heads
    E, mP, Te, T = s.E, s.mP, s.Te, s.T
    dev = d_e.device
    d = torch.cat([d_e, d_p], 0)
    state = ops.grad_scale_state([d])
    ops.scale_by_state(d, state, 1, out=d)                               # max |d| in [1, 2)
    d16 = engine.to16(d, e.dtype)

    # feed-forward
    dg = ops.linear(d16, e.w2_t)                                         # [T, 4C]
    dw2, db2 = weight_grad(d16, s.g), ops.colsum(d)
    du = ops.geglu_bwd(s.u, dg)
    dw1, db1 = weight_grad(du, s.z), ops.colsum(du)
    dz = ops.linear(du, e.w1_t, out_dtype=torch.float32)
    dy, dg2, dbt2 = ops.layernorm_bwd(s.y, e.ln2.g, dz, e.ln2.eps, dres=d)
    dy16 = engine.to16(dy, e.dtype)

    # attention output projection
    da = ops.linear(dy16, e.wo_t)                                        # [T, C]: gradient of the attention output
    dwo, dbo = weight_grad(dy16, s.a), ops.colsum(dy)

    # the two attentions
    da_e, da_p = da[:Te].view(b, E, Cc), da[Te:].view(b, mP, Cc)
    a_e, a_p = s.a[:Te].view(b, E, Cc), s.a[Te:].view(b, mP, Cc)
    dat_e, dat_p = ops.transpose_tokens(da_e), ops.transpose_tokens(da_p)
    delta_e, delta_p = ops.attention_delta(a_e, da_e, b, H, 32, E), ops.attention_delta(a_p, da_p, b, H, 32, mP)
    qkv_e, qkv_p = s.qkv[:Te].view(b, E, 3 * Cc), s.qkv[Te:].view(b, mP, 3 * Cc)
    dqkv = torch.empty_like(s.qkv)
    dqkv_e, dqkv_p = dqkv[:Te].view(b, E, 3 * Cc), dqkv[Te:].view(b, mP, 3 * Cc)
    ld = 3 * Cc
    q_, k_, v_ = slice(0, Cc), slice(Cc, 2 * Cc), slice(2 * Cc, 3 * Cc)
    for i, n, t in _attention_calls(b, s.shared, tables):
        sl = slice(i, i + n)
        strides = dict(q_ld=ld, k_ld=ld, v_ld=ld, do_ld=Cc, dq_ld=ld, dk_ld=ld, dv_ld=ld)
        ops.attention_bwd(qkv_e[sl, :, q_], qkv_p[sl, :, k_], qkv_p[sl, :, v_], da_e[sl],
                          s.qkvt_e[sl, q_], s.qkvt_p[sl, k_], dat_e[sl], s.lse_e[sl], delta_e[sl],
                          dqkv_e[sl, :, q_], dqkv_p[sl, :, k_], dqkv_p[sl, :, v_], n, H, 32, E, mP,
                          q_bs=E * ld, k_bs=mP * ld, v_bs=mP * ld, do_bs=E * Cc, dq_bs=E * ld, dk_bs=mP * ld, dv_bs=mP * ld,
                          bias=t.bias_e, flags=t.flags_e, **strides)
        ops.attention_bwd(qkv_p[sl, :, q_], qkv_e[sl, :, k_], qkv_e[sl, :, v_], da_p[sl],
                          s.qkvt_p[sl, q_], s.qkvt_e[sl, k_], dat_p[sl], s.lse_p[sl], delta_p[sl],
                          dqkv_p[sl, :, q_], dqkv_e[sl, :, k_], dqkv_e[sl, :, v_], n, H, 32, mP, E,
                          q_bs=mP * ld, k_bs=E * ld, v_bs=E * ld, do_bs=mP * Cc, dq_bs=mP * ld, dk_bs=E * ld, dv_bs=E * ld,
                          bias=t.bias_p, flags=t.flags_p, **strides)

    # q / k / v projections and the first norm
    dln = ops.linear(dqkv, e.wqkv_t, out_dtype=torch.float32)
    dwqkv = weight_grad(dqkv, s.ln)
    dx = torch.empty(T, Cc, device=dev, dtype=torch.float32)
    dg1 = torch.zeros(Cc, device=dev, dtype=torch.float32)
    dbt1 = torch.zeros_like(dg1)
    for i in range(b):
        t = tables[0] if s.shared else tables[i]
        for r0, r1, pe in ((i * E, (i + 1) * E, t.pe_e), (Te + i * mP, Te + (i + 1) * mP, t.pe_p)):
            _, g_, b_ = ops.layernorm_bwd(s.x[r0:r1], e.ln1.g, dln[r0:r1], e.ln1.eps, pe=pe, dres=dy[r0:r1], dx=dx[r0:r1])
            dg1, dbt1 = ops.add(dg1, g_), ops.add(dbt1, b_)

    grads = [dg1, dbt1, dwqkv[q_], dwqkv[k_], dwqkv[v_], dwo, dbo, dw1, db1, dw2, db2, dg2, dbt2]
    grads = [ops.scale_by_state(g_.contiguous(), state, 2) for g_ in grads]
    ops.scale_by_state(dx, state, 2, out=dx)
    return dx[:Te], dx[Te:], grads


class WarpAttnFunction(torch.autograd.Function):
    """``WarpAttn.forward`` under autograd: inference kernels forward, ``epa_backward`` backward."""

    @staticmethod
    def forward(ctx, module, cameras, pers_x, equi_x, *params):
        with torch.no_grad():
            out_p, out_e = module.forward_inference(pers_x, equi_x, cameras)
        ctx.module, ctx.cameras = module, cameras
        ctx.save_for_backward(pers_x, equi_x)
        return out_p, out_e

    @staticmethod
    def backward(ctx, d_p, d_e):
        pers_x, equi_x = ctx.saved_tensors
        dx_p, dx_e, grads = ctx.module.backward_block(pers_x, equi_x, ctx.cameras, d_p, d_e)
        return (None, None, dx_p, dx_e, *grads)
