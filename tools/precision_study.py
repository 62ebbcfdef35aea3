"""Error budget of 16-bit storage schemes for the dual-branch denoiser, emulated on the CPU oracle.

    python tools/precision_study.py [--width 320] [--lat 32] [--m 2] [--fmt fp16] [--schemes S0,S1,F,...]

The oracle (oracle/mvgen.py, fp32) is run once as the truth; then the same weights run again with
the oracle's layer forwards replaced by versions that round tensors exactly where a storage scheme
of the HIP path rounds them (MFMA operands, weights, branch-internal tensors, residual stream).
Prints the rel-L2 of the two epsilon outputs per scheme and, with --trace, the rel-L2 after every
block (the error budget as a table).  Test infrastructure / design tool: nothing here is imported
by the product.

Rounding classes
  op      MFMA A-operands: GN/LN(+SiLU) outputs, q/k/v, attention probabilities, attention output,
          GEGLU output                                    (16 bit in every single-pass scheme)
  w       GEMM / conv weights
  mid     GEMM outputs that feed a norm, not an MFMA directly: resnet conv1 output (+temb)
  tstream token stream inside a transformer block (proj_in output and the three residual sums)
  stream  block outputs: conv_in, resnet, transformer, down/upsample, EPA outputs
"""
import argparse
import contextlib
import copy
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import geometry as G          # noqa: E402
from oracle import mvgen as MV            # noqa: E402
from oracle import sd2_unet as U          # noqa: E402
from oracle import third_party as tp      # noqa: E402

CLASSES = ("op", "w", "mid", "tstream", "stream")


class Policy:
    def __init__(self, fmt, rounded, split_w=False):
        self.dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[fmt]
        self.rounded = set(rounded)
        self.trace = None
        self.window = None          # set of driver-block indices to round in (sensitivity runs); None: everywhere
        self.block = 0
        self.live_w = False         # round weights on the fly (needed with a window)
        self.tags = None            # restrict rounding to these tags (None: all)
        self.cur = None             # tag of the GEMM being executed (for live weight rounding)
        self.count = None
        self.exact = set()          # tags computed exactly (split-precision / fp32 layers)
        self.exact_cls = None       # None: operands AND weights of an exact tag are exact; {"op"} / {"w"}: only that side (2-pass)
        self.exact_levels = {}      # tag -> set of UNet levels (0 = full resolution .. 3) where it is exact; absent: all
        self.level = 0

    @property
    def active(self):
        return self.window is None or self.block in self.window

    def r(self, cls, x, tag=None):
        if cls in self.rounded and self.active and (self.tags is None or tag in self.tags) and not self.is_exact(tag, cls):
            if self.count is not None:
                self.count[(cls, tag)] = self.count.get((cls, tag), 0) + 1
            return x.to(self.dt).float()
        return x

    def is_exact(self, tag, cls=None):
        if tag not in self.exact:
            return False
        if self.exact_cls is not None and cls is not None and cls not in self.exact_cls and cls in ("op", "w"):
            return False
        lv = self.exact_levels.get(tag)
        return lv is None or self.level in lv

    def mark(self, name, *tensors):
        if self.trace is not None:
            self.trace.append((name, [t.detach().clone() for t in tensors]))


POL = None


def gemm(tag, mod, x):
    """One MFMA GEMM: 16-bit A operand, (live) 16-bit weights -- both attributed to `tag`."""
    p = POL
    p.cur = tag
    y = mod(p.r("op", x, tag))
    p.cur = None
    return y


def resnet_forward(self, x, temb):
    p = POL
    h = gemm("res.conv1", self.conv1, F.silu(self.norm1(x))) + self.time_emb_proj(F.silu(temb))[:, :, None, None]
    h = p.r("mid", h, "res.h")
    sc = x
    if self.conv_shortcut is not None:
        sc = p.r("mid", gemm("res.short", self.conv_shortcut, x), "res.short")   # 1x1 GEMM on the raw stream tensor
    return p.r("stream", sc + gemm("res.conv2", self.conv2, F.silu(self.norm2(h))), "res.out")


def _softmax_pv(p, s, v, tag):
    # flash kernel: P = exp(s - max) rounded to 16 bit for the PV MFMA, row sum in fp32 of the UNROUNDED values
    e = torch.exp(s - s.amax(-1, keepdim=True))
    return torch.bmm(p.r("op", e, tag + ".P"), v) / e.sum(-1, keepdim=True)


def attention_forward(self, x, context=None):
    p = POL
    name = "attn1" if context is None else "attn2"
    context = x if context is None else context
    b, n, _ = x.shape
    h = self.heads

    def split(t):
        return t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)

    q = split(p.r("op", gemm(name + ".to_q", self.to_q, x), name + ".q"))
    k = split(p.r("op", gemm(name + ".to_k", self.to_k, context), name + ".k"))
    v = split(p.r("op", gemm(name + ".to_v", self.to_v, context), name + ".v"))
    s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype), q, k.transpose(-1, -2),
                      beta=0, alpha=self.scale)
    o = _softmax_pv(p, s, v, name)
    o = o.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
    return gemm(name + ".to_out", self.to_out[0], o)


def geglu_forward(self, x):
    a, gate = gemm("ff1", self.proj, x).chunk(2, dim=-1)
    return a * F.gelu(gate)


def ff_forward(self, x):
    return gemm("ff2", self.net[2], self.net[0](x))


def block_forward(self, x, encoder_hidden_states):
    p = POL
    x = p.r("tstream", self.attn1(self.norm1(x)) + x, "t1")
    x = p.r("tstream", self.attn2(self.norm2(x), encoder_hidden_states) + x, "t2")
    return p.r("tstream", self.ff(self.norm3(x)) + x, "t3")


def t2d_forward(self, x, encoder_hidden_states=None):
    p = POL
    b, c, h, w = x.shape
    t = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = p.r("tstream", gemm("proj_in", self.proj_in, t), "t0")
    for blk in self.transformer_blocks:
        t = blk(t, encoder_hidden_states)
    t = gemm("proj_out", self.proj_out, t)
    return U._Sample(p.r("stream", t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x, "tf.out"))


def down_forward(self, x):
    return POL.r("stream", gemm("down", self.conv, x), "down.out")


def up_forward(self, x):
    return POL.r("stream", gemm("up", self.conv, F.interpolate(x, scale_factor=2.0, mode="nearest")), "up.out")


def epa_attn_forward(self, x, context, bias):
    p = POL
    b, n, _ = x.shape
    h = self.heads

    def heads_first(t):
        return t.reshape(b, t.shape[1], h, -1).transpose(1, 2).reshape(b * h, t.shape[1], -1)

    q = heads_first(p.r("op", gemm("epa.to_q", self.to_q, x), "epa.q"))
    k = heads_first(p.r("op", gemm("epa.to_k", self.to_k, context), "epa.k"))
    v = heads_first(p.r("op", gemm("epa.to_v", self.to_v, context), "epa.v"))
    s = torch.einsum("bid,bjd->bij", q, k) * q.shape[-1] ** -0.5 + bias.repeat_interleave(h, dim=0)
    o = _softmax_pv(p, s, v, "epa")
    o = o.reshape(b, h, n, -1).transpose(1, 2).reshape(b, n, -1)
    return gemm("epa.to_out", self.to_out, o)


def epa_tf_forward(self, x, context, bias, query_pe):
    p = POL
    x = p.r("tstream", self.attn1(self.norm1(x + query_pe), self.norm1(context), bias) + x, "epa.t1")
    return p.r("stream", self.ff(self.norm2(x)) + x, "epa.out")


def epa_geglu_forward(self, x):
    a, gate = gemm("epa.ff1", self.proj, x).chunk(2, dim=-1)
    return a * F.gelu(gate)


def epa_ff_forward(self, x):
    return gemm("epa.ff2", self.net[2], self.net[0](x))


PATCHES = [(U.ResnetBlock2D, resnet_forward), (U.Attention, attention_forward), (U.GEGLU, geglu_forward),
           (U.BasicTransformerBlock, block_forward), (U.Transformer2DModel, t2d_forward),
           (U.Downsample2D, down_forward), (U.Upsample2D, up_forward),
           (MV._BiasedCrossAttention, epa_attn_forward), (MV._EPATransformer, epa_tf_forward),
           (MV._GEGLU, epa_geglu_forward), (U.FeedForward, ff_forward), (MV._FeedForward, epa_ff_forward)]


@contextlib.contextmanager
def emulate(policy):
    global POL
    saved = [(c, c.forward) for c, _ in PATCHES]
    for c, f in PATCHES:
        c.forward = f
    POL = policy
    # block-level trace + rounding of the tensors the branch driver itself produces (conv_in, head input)
    orig = {n: getattr(MV._Branch, n) for n in ("resnet", "attention", "downsample", "upsample", "__init__", "head")}

    def wrap(name):
        def f(self, *a, **k):
            policy.block += 1
            policy.names[policy.block] = "%s.%s" % ("pano" if getattr(self, "pano", a and len(a) > 4 and a[4]) else "pers", name.strip("_"))
            if name == "__init__":
                self._level = 0
            policy.level = self._level - (1 if name == "upsample" else 0)     # an upsampling conv works at the finer level
            out = orig[name](self, *a, **k)
            if name == "downsample":
                self._level += 1
            elif name == "upsample":
                self._level -= 1
            if name == "__init__":
                self.h = policy.r("stream", self.h, "conv_in")
                self.skips = [self.h]
                self._tag = "pano" if self.pano else "pers"
                self._n = 0
            self._n += 1
            policy.mark("%s.%02d.%s" % (self._tag, self._n, name.strip("_")), self.h)
            return out
        return f

    def head(self):
        y = policy.r("op", self.u.conv_act(self.u.conv_norm_out(self.h)), "head")
        return self._wrapped(self.u.conv_out, y, 1, 1)

    def head_counted(self):
        policy.block += 1
        policy.names[policy.block] = "%s.head" % ("pano" if self.pano else "pers")
        policy.level = 0
        return head(self)

    for n in ("resnet", "attention", "downsample", "upsample", "__init__"):
        setattr(MV._Branch, n, wrap(n))
    MV._Branch.head = head_counted
    epa_orig = MV.EPABlock.forward

    def epa_counted(self, *a, **k):
        policy.block += 1
        policy.names[policy.block] = "epa"
        return epa_orig(self, *a, **k)

    MV.EPABlock.forward = epa_counted
    lin_orig, conv_orig = F.linear, F.conv2d

    def lin_live(x, w, b=None):
        return lin_orig(x, policy.r("w", w, policy.cur) if policy.live_w and policy.cur else w, b)

    def conv_live(x, w, b=None, *a, **k):
        return conv_orig(x, policy.r("w", w, policy.cur) if policy.live_w and policy.cur else w, b, *a, **k)

    F.linear, F.conv2d = lin_live, conv_live
    policy.block = 0
    policy.names = {}
    try:
        yield
    finally:
        F.linear, F.conv2d = lin_orig, conv_orig
        MV.EPABlock.forward = epa_orig
        for c, f in saved:
            c.forward = f
        for n, f in orig.items():
            setattr(MV._Branch, n, f)
        POL = None


@torch.no_grad()
def fold_lora(model):
    for mod in model.modules():
        if isinstance(mod, U.LoRACompatibleLinear) and mod.lora_layer is not None:
            mod.weight += mod.lora_layer.up.weight @ mod.lora_layer.down.weight
            mod.lora_layer = None


def build(width, ctx, seed=11, heads=None):
    if width == 320:
        cfg = dict(U.SD2_BASE)
    else:
        cfg = U.tiny_config(width=width, cross_attention_dim=ctx, heads=heads or (1, 2, 4, 4), groups=32)
    unet, pano_unet = U.UNet2DConditionModel(**cfg), U.UNet2DConditionModel(**cfg)
    unet.add_lora(4)
    pano_unet.add_lora(4)
    U.init_synthetic(unet, seed)
    U.init_synthetic(pano_unet, seed + 1)
    model = MV.DualBranchDenoiser(unet, pano_unet, None, None, True)
    U.init_synthetic(model.cp_blocks_encoder, seed + 2)
    U.init_synthetic(model.cp_blocks_mid, seed + 3)
    U.init_synthetic(model.cp_blocks_decoder, seed + 4)
    MV.randomize_epa(model, seed + 5)
    fold_lora(model)
    return model, cfg


SCHEMES = {
    "S0": CLASSES,                                  # everything 16 bit (round-1 path)
    "S1": ("op", "w", "mid"),                       # fp32 residual + token streams
    "S1b": ("op", "w", "mid", "tstream"),           # fp32 block outputs only
    "F": ("op", "w"),                               # floor of single-pass 16-bit MFMA: operands only
    "only_op": ("op",), "only_w": ("w",), "only_mid": ("mid",), "only_tstream": ("tstream",),
    "only_stream": ("stream",),
}


EXACT = {
    "sp": {"res.short", "proj_in", "proj_out"},
    "du": {"down", "up", "head"}, "d": {"down"}, "u": {"up"}, "hd": {"head"},
    "s": {"res.short"}, "pi": {"proj_in"}, "po": {"proj_out"},
    "h": {"res.h"},
    "c2": {"res.conv2"}, "c1": {"res.conv1"},
    "ff": {"ff1", "ff2"},
    "epa": {"epa.to_q", "epa.to_k", "epa.to_v", "epa.to_out", "epa.ff1", "epa.ff2", "epa.q", "epa.k", "epa.v", "epa.P"},
}


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--lat", type=int, default=32)
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--fmt", default="fp16")
    ap.add_argument("--schemes", default="S0,S1,S1b,F,only_op,only_w,only_mid,only_tstream,only_stream")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--tagsens", default=None, help="per-tag sensitivity inside these driver blocks: all | 3-7,84-91")
    ap.add_argument("--sens", default=None, help="per-block sensitivity: round the classes of this scheme inside ONE driver block at a time")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    t0 = time.time()
    model, cfg = build(args.width, args.ctx)
    ctx = cfg["cross_attention_dim"]
    g = torch.Generator().manual_seed(0)
    b, m, L = 1, args.m, 77
    lat = torch.randn(b, m, 4, args.lat, args.lat, generator=g)
    pl = torch.randn(b, 1, 4, args.lat, 2 * args.lat, generator=g)
    pe, ppe = torch.randn(b, m, L, ctx, generator=g), torch.randn(b, 1, L, ctx, generator=g)
    t = torch.full((b, m), 981, dtype=torch.long)
    th = [0.0, 36.0, 180.0, 270.0][:m]
    ph = [0.0, 52.6, -10.8, 10.8][:m]
    cams = {"FoV": torch.full((b, m), 90), "theta": torch.tensor([th] * b, dtype=torch.float64),
            "phi": torch.tensor([ph] * b, dtype=torch.float64)}
    print("built in %.1f s" % (time.time() - t0), flush=True)
    truth_pol = Policy("fp16", ())
    truth_pol.trace = [] if args.trace else None
    with torch.no_grad(), emulate(truth_pol):
        t0 = time.time()
        ws, wp = model(lat, pl, t, pe, ppe, cams)
        print("fp32 truth in %.1f s" % (time.time() - t0), flush=True)
    def run(pol):
        pol.live_w = True
        with torch.no_grad(), emulate(pol):
            s, ps = model(lat, pl, t, pe, ppe, cams)
        return rel(s, ws), rel(ps, wp)

    nblocks, names = truth_pol.block, truth_pol.names
    if args.sens:
        rows = []
        for i in range(1, nblocks + 1):
            pol = Policy(args.fmt, SCHEMES[args.sens])
            pol.window = {i}
            rows.append((i, names[i]) + run(pol))
            print("%3d %-18s views %.3e pano %.3e" % rows[-1], flush=True)
        print("root-sum-square over blocks: views %.3e pano %.3e"
              % (sum(r[2] ** 2 for r in rows) ** 0.5, sum(r[3] ** 2 for r in rows) ** 0.5))
        return
    if args.tagsens:
        # per-GEMM / per-tensor sensitivity inside a set of driver blocks ("all" or e.g. "3-7,84-91,99")
        window = None
        if args.tagsens != "all":
            window = set()
            for part in args.tagsens.split(","):
                lo, _, hi = part.partition("-")
                window |= set(range(int(lo), int(hi or lo) + 1))
        pol = Policy(args.fmt, CLASSES)
        pol.window, pol.count = window, {}
        print("all classes in window: views %.3e pano %.3e" % run(pol))
        rows = []
        for (cls, tag) in sorted(pol.count):
            q = Policy(args.fmt, (cls,))
            q.window, q.tags = window, {tag}
            rows.append((cls, tag) + run(q))
            print("%-8s %-16s views %.3e pano %.3e" % rows[-1], flush=True)
        print("root-sum-square: views %.3e pano %.3e"
              % (sum(r[2] ** 2 for r in rows) ** 0.5, sum(r[3] ** 2 for r in rows) ** 0.5))
        return
    for name in args.schemes.split(","):
        base, _, ex = name.partition("+")
        side = None
        if base.endswith(":op") or base.endswith(":w"):
            base, side = base.rsplit(":", 1)
        pol = Policy(args.fmt, SCHEMES[base])
        pol.exact_cls = {side} if side else None
        for e in [e for e in ex.split("+") if e]:
            grp, _, lv = e.partition("@")
            pol.exact |= EXACT[grp]
            if lv:
                for tag in EXACT[grp]:
                    pol.exact_levels[tag] = set(int(c) for c in lv)
        pol.trace = [] if args.trace else None
        es, ep = run(pol)
        print("%-12s %s  views %.3e  pano %.3e" % (name, args.fmt, es, ep), flush=True)
        if args.trace:
            for (n0, a), (n1, b_) in zip(truth_pol.trace, pol.trace):
                print("    %-28s %.3e" % (n0, rel(b_[0], a[0])))


if __name__ == "__main__":
    main()
