"""Parity of the MFMA kernels at the FULL sizes of BASELINE.json configs[1] (40 view samples of 64x64 latents,
SD-2-base widths), where the CPU oracle would take hours: the checker is an independent fp32 statement of the
same op evaluated by PyTorch's own GPU kernels (test infrastructure, like the oracle), on the 16-bit-rounded
inputs.  Plus size-independent properties of the whole step (determinism, graph replay == eager).
Needs an MI355X: `-m gpu`."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {torch.bfloat16: 4e-3, torch.float16: 6e-4}      # one 16-bit rounding of the output (see test_gpu_kernels)


def ops():
    from panfusion_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV) * scale


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["conv64", "conv16_tail_split", "conv8_split_k", "conv32_s2", "up32"])
def test_conv3x3_full_size(dtype, case):
    n, h, w, cin, cout, stride, up = {"conv64": (40, 64, 64, 320, 320, 1, 0), "conv16_tail_split": (40, 16, 16, 1280, 1280, 1, 0),
                                      "conv8_split_k": (40, 8, 8, 1280, 1280, 1, 0), "conv32_s2": (40, 32, 32, 640, 640, 2, 0),
                                      "up32": (40, 32, 32, 640, 640, 1, 1)}[case]
    x = rnd(n, h, w, cin, seed=1).to(dtype)
    wt = (rnd(cout, cin, 3, 3, seed=2) / (9 * cin) ** 0.5).to(dtype)
    b, table = rnd(cout, seed=3), rnd(n, cout, seed=4)
    xin = x.float().permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xin, wt.float(), b, stride=stride, padding=1) + table[:, :, None, None]
    got = ops().conv_gemm(x, wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous(), cout, n_img=n, h_in=h, w_in=w, ksize=3,
                          stride=stride, pad=1, upsample=up, bias=b, rowvec=table)
    got = got.view(n, want.shape[2], want.shape[3], cout).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert rel(got, want) <= TOL[dtype], (case, rel(got, want))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_ff_geglu_and_residual_linear_full_size(dtype):
    """The FF pair of the C=320 transformer blocks at 40 x 4096 tokens: fused-GEGLU projection, then the K=1280
    projection with bias + residual."""
    o = ops()
    M, Cc = 40 * 4096, 320
    x = rnd(M, Cc, seed=5).to(dtype)
    w1 = (rnd(8 * Cc, Cc, seed=6) / Cc ** 0.5).to(dtype)
    b1 = rnd(8 * Cc, seed=7)
    w1i, b1i = o.interleave_geglu(w1, b1)
    y = x.float() @ w1.float().T + b1
    want1 = y[:, :4 * Cc] * F.gelu(y[:, 4 * Cc:])
    got1 = o.linear(x, w1i, bias=b1i, geglu=True)
    assert rel(got1, want1) <= TOL[dtype], rel(got1, want1)
    w2 = (rnd(Cc, 4 * Cc, seed=8) / (4 * Cc) ** 0.5).to(dtype)
    b2 = rnd(Cc, seed=9)
    want2 = got1.float() @ w2.float().T + b2 + x.float()
    got2 = o.linear(got1, w2, bias=b2, residual=x)
    assert rel(got2, want2) <= TOL[dtype], rel(got2, want2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_self_attention_full_size(dtype):
    """UNet self-attention at 64x64: 40 samples x 5 heads x 4096^2, head dim 64 (fused q|k buffer, transposed V)."""
    o = ops()
    B, H, D, N = 40, 5, 64, 4096
    Cc = H * D
    qk = rnd(B * N, 2 * Cc, seed=10).to(dtype)
    vt = rnd(B, Cc, N, seed=11).to(dtype)
    got = o.attention(qk, qk[:, Cc:], vt, B, H, D, N, N, q_ld=2 * Cc, k_ld=2 * Cc, vt_ld=N, q_bs=N * 2 * Cc,
                      k_bs=N * 2 * Cc, vt_bs=Cc * N)
    q = qk[:, :Cc].float().view(B, N, H, D).transpose(1, 2)
    k = qk[:, Cc:].float().view(B, N, H, D).transpose(1, 2)
    v = vt.float().view(B, H, D, N).transpose(2, 3)
    want = torch.empty(B, N, Cc, device=DEV)
    for b0 in range(0, B, 8):                                      # 8 samples at a time: 2.7 GB of fp32 scores
        s = torch.softmax(q[b0:b0 + 8] @ k[b0:b0 + 8].transpose(-1, -2) * D ** -0.5, -1)
        want[b0:b0 + 8] = (s @ v[b0:b0 + 8]).transpose(1, 2).reshape(-1, N, Cc)
    assert torch.isfinite(got).all()
    assert rel(got, want) <= 2 * TOL[dtype], rel(got, want)       # P is rounded to 16 bit before P.V as well


def test_epa_attention_full_size_with_bias_tables():
    """EPA panorama-query direction at s=2 of the benchmark geometry: 2048 queries x 20480 keys x 20 heads of 32, the
    real bias / flag tables of the 20 icosahedron cameras (98-99 % of the 32x32 tiles are skipped)."""
    from oracle import geometry as G
    import numpy as np
    o = ops()
    th, ph = G.icosahedron_cameras()
    bias_e, bias_p, flags_e, flags_p = o.epa_tables([90] * 20, np.degrees(th), np.degrees(ph), 32, 32, 32, 64, DEV)
    B, H, D, E, mP = 2, 20, 32, 2048, 20480
    Cc = H * D
    dtype = torch.float16
    q = rnd(B * E, Cc, seed=12).to(dtype)
    k = rnd(B * mP, Cc, seed=13).to(dtype)
    vt = rnd(B, Cc, mP, seed=14).to(dtype)
    got = o.attention(q, k, vt, B, H, D, E, mP, q_ld=Cc, k_ld=Cc, vt_ld=mP, q_bs=E * Cc, k_bs=mP * Cc, vt_bs=Cc * mP,
                      bias=bias_e, flags=flags_e)
    qh = q.float().view(B, E, H, D).transpose(1, 2)
    kh = k.float().view(B, mP, H, D).transpose(1, 2)
    vh = vt.float().view(B, H, D, mP).transpose(2, 3)
    want = torch.empty(B, E, Cc, device=DEV)
    for b0 in range(B):
        s = torch.softmax(qh[b0] @ kh[b0].transpose(-1, -2) * D ** -0.5 + bias_e[None], -1)
        want[b0] = (s @ vh[b0]).transpose(0, 1).reshape(E, Cc)
    assert rel(got, want) <= 2 * TOL[dtype], rel(got, want)


def test_full_size_step_is_deterministic_and_graph_replay_equals_eager():
    """BASELINE.json configs[1] shapes end to end (random-init SD-2-base-shaped weights): two eager denoiser calls are
    bit-identical (no atomics, split-K slabs summed in order, two streams joined at the EPA blocks) and three graph
    replayed loop steps equal three eager ones bit for bit."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from panfusion_amd.models.sd2_unet_params import SD2_BASE
    from panfusion_amd.pipeline import DenoiseLoop
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    cfg = dict(SD2_BASE)
    th, ph = icosahedron_sample_camera()
    model = bench.build_model(torch.device(DEV), torch.bfloat16, cfg)
    inputs = bench.build_inputs(torch.device(DEV), 20, (64, 64), (64, 128), cfg["cross_attention_dim"], (np.degrees(th), np.degrees(ph)))
    outs = []
    for graphs in (False, True):
        loop = DenoiseLoop(model, *inputs, steps=3, use_graphs=graphs)
        if not graphs:
            a = loop._denoise(loop.cameras)
            b = loop._denoise(loop.cameras)
            assert all(torch.equal(x, y) for x, y in zip(a, b))
            assert all(torch.isfinite(x).all() for x in a)
        outs.append([x.clone() for x in loop.run()])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ---- size-independent properties at the BASELINE.json configs[1] sizes (no oracle can run these shapes in seconds) ----
def test_conv_scaling_by_powers_of_two_is_exact_full_size():
    """Scaling the input of the implicit-GEMM conv by 2^k scales every product and partial sum by 2^k exactly, so the
    fp32-accumulated output scales bit for bit (64^2 level: 40 images, 320 -> 320, K = 2880); the split-precision 1x1
    GEMM of the mixed scheme on an fp32 stream tensor scales to 1e-6, and additivity holds to one rounding."""
    from panfusion_amd import engine
    o = ops()
    n, h, w, cin, cout = 40, 64, 64, 320, 320
    keep_normal = lambda t: torch.where(t.abs() < 2 ** -9, torch.full_like(t, 2 ** -9), t)   # no fp16 subnormals in or near the operands
    x = keep_normal(rnd(n, h, w, cin, seed=21)).half()
    wt = keep_normal(rnd(cout, 9 * cin, seed=22) / (9 * cin) ** 0.5 * 16).half()
    run = lambda t: o.conv_gemm(t, wt, cout, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, out_dtype=torch.float32)
    y = run(x)
    assert torch.equal(run(x * 4.0), y * 4.0) and torch.equal(run(x * 32.0), y * 32.0)
    x2 = rnd(n, h, w, cin, seed=23).half()
    s = run(x + x2).float()                       # x + x2 is rounded to fp16 first: compare against the same operand
    want = run(x).float() + run(x2).float()
    assert rel(s, want) <= 8e-4
    xs = keep_normal(rnd(n * h * w, cin, seed=24))             # fp32 stream tensor -> [hi | lo] pair -> 3-pass GEMM, fp32 out
    w3 = engine._split_weight(rnd(cout, cin, seed=25) / cin ** 0.5, 1, DEV, torch.float16)
    ex = lambda t: engine.exact_gemm(engine.split_operand(t, dtype=torch.float16), w3, cout, w_in=n * h * w, out_dtype=torch.float32)
    # (not bit for bit: the lo halves of small activations, ~2^-12 |x|, sit in fp16's subnormal range, where a scaling
    # changes their rounding -- an absolute error of <= 2^-25 per element)
    assert rel(ex(xs * 8.0), ex(xs) * 8.0) <= 1e-6


def test_attention_shift_and_permutation_invariance_full_size():
    """softmax is invariant to a constant added to a whole score row (here: to every entry of the EPA bias table, all tiles
    flagged) and attention is invariant to a permutation of the keys applied to K rows, V^T columns and bias columns alike:
    checked on the panorama-query EPA direction at the benchmark geometry (2048 queries x 20480 keys x 20 heads of 32)."""
    o = ops()
    B, H, D, E, mP = 1, 20, 32, 2048, 20480
    Cc = H * D
    q = rnd(B * E, Cc, seed=31).half()
    k = rnd(B * mP, Cc, seed=32).half()
    vt = rnd(B, Cc, mP, seed=33).half()
    bias = (rnd(E, mP, seed=34) * 0.5).contiguous()
    flags = torch.ones(E // 32, mP // 32, dtype=torch.uint8, device=DEV)
    run = lambda kk, vv, bb: o.attention(q, kk, vv, B, H, D, E, mP, q_ld=Cc, k_ld=Cc, vt_ld=mP, q_bs=E * Cc, k_bs=mP * Cc,
                                         vt_bs=Cc * mP, bias=bb, flags=flags)
    base = run(k, vt, bias)
    shifted = run(k, vt, bias + 3.0)
    assert rel(shifted, base) <= 1.5e-3           # two independent fp16 roundings of P and O
    perm = torch.randperm(mP, generator=torch.Generator(device=DEV).manual_seed(35), device=DEV)
    permuted = run(k.view(B, mP, Cc)[:, perm].reshape(B * mP, Cc).contiguous(), vt[:, :, perm].contiguous(), bias[:, perm].contiguous())
    assert rel(permuted, base) <= 1.5e-3
    assert torch.isfinite(base).all()
