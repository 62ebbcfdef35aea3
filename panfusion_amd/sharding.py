"""Multi-GPU sharding of one denoising step (one process per GPU, ``torch.distributed`` over
RCCL/xGMI on the GPU box, gloo in the CPU tests).

The reference has no intra-step parallelism (views are a batch dimension on one GPU, SURVEY.md
§2.1); this is new design.  Independent units of a step:
  * the two classifier-free-guidance samples (independent everywhere, EPA included);
  * the m perspective views (independent everywhere except inside EPA).
Layout for N ranks (N even): 2 CFG halves x G = N/2 view groups; rank r -> (c, g) = (r // G, r % G),
views [g*m/G, (g+1)*m/G) ("even"), or -- from G >= 4 -- the "pano_rank" layout of plan(): group 0 of a half owns
the panorama branch and fewer views, the others run the view branch only.  N = 2 is the pure CFG split: no traffic
inside the step.

Exchanges per step:
  * per EPA block (7x), only when G > 1: ONE all-gather inside the CFG half of the layer-normed view
    tokens LN1(x_p + PE) (m/G * P * C 16-bit per rank; <= 3.3 MB at s=2, C=320, G=4).  Every rank then
    projects K and V^T for all views locally (the K/V projections are ~8 GFLOP per block -- cheaper
    than moving 2C-wide K|V) and runs the panorama-query direction redundantly; the view-query
    direction needs no traffic because the panorama branch is replicated inside a CFG half.
  * once per step: ONE all-gather of the epsilon predictions (1.3 MB views + 0.13 MB pano), after
    which every rank applies the CFG merge + DDIM update to its replica of the latents.
All collectives are small and latency bound on xGMI; none is a translation of a reference call.
"""
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist

from . import ops
from .pipeline import DenoiseLoop


@dataclass
class ShardInfo:
    rank: int
    world: int
    cfg: int            # which CFG sample this rank computes (0 = uncond, 1 = cond)
    g: int              # view-group index inside the CFG half
    G: int              # view groups per CFG half
    m: int              # total number of views
    group: object = None      # process group of the CFG half (None when G == 1)
    split: tuple = None       # views per group (None: m / G each)
    pano_g: int = None        # the group that computes the panorama branch (None: every rank does, replicated)

    @property
    def counts(self):
        return tuple(self.split) if self.split is not None else (self.m // self.G,) * self.G

    @property
    def views(self):
        c = self.counts
        v0 = sum(c[:self.g])
        return v0, v0 + c[self.g]

    @property
    def vmax(self):
        return max(self.counts)

    @property
    def has_pano(self):
        return self.pano_g is None or self.pano_g == self.g

    @property
    def has_views(self):
        """False on a panorama-only owner (explicit split with 0 views for group 0)."""
        return self.counts[self.g] > 0

    @property
    def pano_src(self):
        """Global rank of the panorama owner of this CFG half."""
        return self.cfg * self.G + (self.pano_g or 0)


# TIME model of one rank's step on one MI355X, per configuration (panorama latent, view latent): a view-only rank takes
# base + per_view * views, the panorama owner WITH views base + per_view * views + pano, an owner WITHOUT views pano_only
# (cheaper than base + pano: no second chain of latency-bound kernels competing with the panorama branch's).  Measured with
# tools/sim_rank.py (the sharded loop of one rank in a single process, collectives replaced by local stand-ins; fp16 mixed;
# profiles/r4_sim_ranks.txt).  By FLOPs the cfg-2 panorama branch is only 2.7 views (1.918 / 0.804 TFLOP + EPA);
# in time it is ~8: two samples through layers too small to fill 256 CUs.  At cfg 4 (128 x 256 panorama latent) the
# panorama branch IS the step: an owner without views is 3x slower than a 7-view rank, whatever the split.
TIME_MODEL = {
    # (pano_hw, lat_hw, layout_cond): milliseconds; `pano` / `pano_only` are what the panorama branch adds ABOVE `base` on an owner
    # with / without views.  Round 6: re-measured at HEAD by tools/fit_time_model.py, raw timings and fit in
    # profiles/r6_time_model.json (final build: packed conv offsets, EPA kernel at four waves per SIMD with its key range split on the owner) -- cfg 2: 7-view
    # rank 13.72, 14-view rank 19.65, owner alone 10.53, owner with 6 views 18.12; cfg 5: 13.71 / 19.68 / 14.91 / 22.81; cfg 4 (query split off): 14.21 / 20.46 /
    # 22.82 / 30.86.  (The first fit of the round, two builds earlier: 14.19 / 20.16 / 11.84 / 19.53 -- the owner's EPA attentions were 0.6 ms slower.)  (Round 4's constants -- base 7.75, per_view 0.956,
    # pano 6.5 / 4.0 -- were still in here through round 5 while the kernels under them changed: the per-view cost fell by 11 %, the base rose.)
    ((64, 128), (64, 64), False): dict(base=7.79, per_view=0.847, pano=5.25, pano_only=2.74),       # cfg 2 / 3
    ((64, 128), (64, 64), True): dict(base=7.74, per_view=0.853, pano=9.95, pano_only=7.17),        # cfg 5: + the panorama ControlNet
    # cfg 4; the query split of the five 32 768-token self-attentions of the panorama branch (split_pano_attention), measured at 4 ranks per
    # half: the owner sheds 3.13 ms (22.82 -> 19.69: attn (1 - 1 / G) with attn = 4.17 -- a quarter of the rows is 320 workgroups on 256 CUs and
    # costs 0.4 of the whole attention, and every attention adds two graph-segment breaks), every other rank takes on 2.94 ms
    # (14.21 -> 17.15: attn_help / G with attn_help = 11.76).  Round 5 carried ONE constant (11 ms) for both sides and overstated what the owner sheds.
    ((128, 256), (64, 64), False): dict(base=7.96, per_view=0.893, pano=17.54, pano_only=14.86, attn=4.17, attn_help=11.76),
}
_DEFAULT_KEY = ((64, 128), (64, 64), False)
_WARNED = set()


def time_model(pano_hw=None, lat_hw=None, layout_cond=False):
    """The constants for a configuration; an unmeasured one scales the cfg-2 panorama terms by the panorama token count (the
    self-attention term grows faster: an underestimate, flagged by `measured=False`)."""
    key = (tuple(pano_hw), tuple(lat_hw), bool(layout_cond)) if pano_hw is not None and lat_hw is not None else _DEFAULT_KEY
    if key in TIME_MODEL:
        return dict(TIME_MODEL[key], measured=True, pano_tokens=key[0][0] * key[0][1])
    # unmeasured: scale the cfg-2 row that matches the layout condition (a ControlNet rides on the owner: ADVICE r4) and say so once
    ref = TIME_MODEL[(_DEFAULT_KEY[0], _DEFAULT_KEY[1], key[2])]
    if key not in _WARNED:
        _WARNED.add(key)
        if not dist.is_initialized() or dist.get_rank() == 0:
            import warnings
            warnings.warn("panfusion_amd.sharding: no measured time model for panorama latent %s / view latent %s / layout_cond=%s -- the "
                          "view split scales the cfg-2 constants by token count (the panorama self-attention grows faster: the owner may "
                          "carry too many views); pass split=... or PF_SHARD_SPLIT to override" % (key[0], key[1], key[2]))
    r_p = (key[0][0] * key[0][1]) / (64.0 * 128.0)
    r_v = (key[1][0] * key[1][1]) / (64.0 * 64.0)
    return dict(base=ref["base"], per_view=ref["per_view"] * r_v, pano=ref["pano"] * r_p, pano_only=ref["pano_only"] * r_p, measured=False,
                pano_tokens=key[0][0] * key[0][1])


def _split_rule(G, tokens):
    """Is a panorama self-attention of `tokens` tokens query-split over a CFG half of G ranks?  The ONE predicate of the planner
    (_attn_shares) and of the run time (splits_pano_attention): ADVICE r5 -- with G = 3 or 5 the 32-row granularity fails and the
    planner must not book a saving that does not happen."""
    return G is not None and G >= max(2, ATTN_SPLIT_MIN_GROUP) and tokens >= ATTN_SPLIT_MIN_TOKENS and tokens % (32 * G) == 0


def _attn_shares(tm, G):
    """(ms the owner sheds, ms every other rank of the half takes on) under the self-attention query split."""
    a = tm.get("attn", 0.0)
    if not a or not _split_rule(G, tm.get("pano_tokens", 0)):
        return 0.0, 0.0
    return a * (1.0 - 1.0 / G), tm.get("attn_help", a) / G


def owner_cost(m0, tm=None, G=None):
    """Step time of the panorama owner holding m0 views, in view units (per_view = 1) above a view-only rank's base."""
    tm = tm or time_model()
    shed, _ = _attn_shares(tm, G)
    return ((tm["pano_only"] if m0 == 0 else m0 * tm["per_view"] + tm["pano"]) - shed) / tm["per_view"]


def helper_cost(views, tm=None, G=None):
    """A view-only rank of the panorama-rank layout, in view units: its views + its share of the split self-attentions."""
    tm = tm or time_model()
    _, extra = _attn_shares(tm, G)
    return views + extra / tm["per_view"]


def pano_rank_split(m, G, tm=None):
    """Views per group when group 0 owns the panorama branch: (m0, ...rest spread as evenly as possible) for
    the m0 >= 0 that minimises the slowest group, max(owner_cost(m0), ceil((m - m0) / (G - 1))); m0 = 0 is a
    panorama-only owner.  None if G < 2 or the views do not give every other group at least one."""
    if G < 2 or m < G - 1:
        return None
    best, best_cost = None, None
    for m0 in range(0, m - (G - 1) + 1):
        rest = m - m0
        cost = max(owner_cost(m0, tm, G), helper_cost(-(-rest // (G - 1)), tm, G))
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = m0, cost
    rest, q, r = m - best, (m - best) // (G - 1), (m - best) % (G - 1)
    return (best,) + tuple(q + (1 if i < r else 0) for i in range(G - 1))


def split_cost(split, pano_replicated, tm=None):
    """Slowest group of a layout in view units (the model above)."""
    tm = tm or time_model()
    if pano_replicated:
        return max(split) + tm["pano"] / tm["per_view"]
    return max(owner_cost(split[0], tm, len(split)), helper_cost(max(split[1:]), tm, len(split)))


def step_time_ms(split, pano_replicated, tm=None):
    """Predicted step time (ms) of the slowest rank of a layout: what tools/sim_rank.py measures."""
    tm = tm or time_model()
    return tm["base"] + split_cost(split, pano_replicated, tm) * tm["per_view"]


def plan(world, rank, m, layout="auto", split=None, pano_hw=None, lat_hw=None, layout_cond=False):
    """Layout of `rank`: 2 CFG halves x G = world/2 view groups.
    layout "even": m/G views per group, the panorama branch replicated inside a CFG half.  layout "pano_rank"
    (chosen by "auto" whenever its slowest group is faster by the time model above, i.e. from G >= 2): group 0
    of a half owns the panorama branch and fewer views (7 / 13 for m = 20, G = 2; none at all, 0 / 7 / 7 / 6, for G = 4); the
    other groups run the view branch only and receive the layer-normed panorama tokens by a broadcast at every
    EPA block.  PF_SHARD_SPLIT=a,b,... overrides the split."""
    if world < 2 or world % 2:
        raise ValueError("sharded step needs an even number of ranks (CFG pair x view groups), got %d" % world)
    G = world // 2
    info = ShardInfo(rank=rank, world=world, cfg=rank // G, g=rank % G, G=G, m=m)
    if split is not None:
        # group 0 (the panorama owner) may be given no views at all: it then runs the panorama branch and the
        # panorama-query half of every EPA block only
        if len(split) != G or sum(split) != m or min(split) < 0 or (G > 1 and min(split[1:]) < 1) or (G == 1 and split[0] < 1):
            raise ValueError("split %r does not distribute %d views over %d groups" % (split, m, G))
        info.split, info.pano_g = tuple(split), 0
        return info
    tm = time_model(pano_hw, lat_hw, layout_cond)         # per configuration: the panorama branch of cfg 4 costs 6x that of cfg 2
    if layout == "auto":                                  # whichever the time model says is faster
        sp = pano_rank_split(m, G, tm)
        even_ok = m % G == 0
        if sp is not None and (not even_ok or split_cost(sp, False, tm) < split_cost((m // G,) * G, True, tm)):
            layout = "pano_rank"
        else:
            layout = "even"
    if layout == "pano_rank":
        sp = pano_rank_split(m, G, tm)
        if sp is None:
            raise ValueError("no panorama-rank split of %d views over %d groups" % (m, G))
        info.split, info.pano_g = sp, 0
        return info
    if m % G:
        raise ValueError("%d views do not divide over %d view groups" % (m, G))
    return info


def make_shard(m, layout=None, split=None, pano_hw=None, lat_hw=None, layout_cond=False):
    """Build this process' ShardInfo and the process group of its CFG half (collective call:
    every rank creates every group, in the same order).  pano_hw / lat_hw select the time model of the configuration."""
    import os
    world, rank = dist.get_world_size(), dist.get_rank()
    if split is None and os.environ.get("PF_SHARD_SPLIT"):
        split = tuple(int(v) for v in os.environ["PF_SHARD_SPLIT"].split(","))
    info = plan(world, rank, m, layout or os.environ.get("PF_SHARD_LAYOUT", "auto"), split, pano_hw=pano_hw, lat_hw=lat_hw, layout_cond=layout_cond)
    if info.G > 1:
        groups = [dist.new_group(list(range(c * info.G, (c + 1) * info.G))) for c in range(2)]
        info.group = groups[info.cfg]
    return info


class SegmentedGraph:
    """A launch sequence recorded as hipGraph segments separated by eager calls.

    The denoiser of a view-sharded rank contains 7 small collectives (one per EPA block).  Launching
    its ~1100 kernels eagerly costs more host time than the GPU needs at 4-8 ranks, and capturing RCCL
    calls inside a graph ties the graph to the communicator's internals; so the kernel stretches BETWEEN
    collectives are captured (one graph each, one shared memory pool) and the collectives stay ordinary
    eager calls on the same stream, re-issued on the same tensors at every replay.

        seg = SegmentedGraph()
        with seg.record():
            out = fn()              # fn calls seg.eager(callable) (via RECORDER) where it must break
        seg.replay()                # graphs and eager calls in recorded order; `out` holds the results
    """

    def __init__(self):
        self.items = []             # ("graph", CUDAGraph) | ("call", callable)
        self.pool = None
        self._cur = None
        self._stream = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()        # one memory pool shared by all segments
        # thread_local: the RCCL watchdog thread polls events of the eager collectives issued between
        # segments; in the default "global" mode such a call from another thread invalidates the capture
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._cur = g

    def _end(self):
        self._cur.capture_end()
        self.items.append(("graph", self._cur))
        self._cur = None

    def record(self):
        seg = self

        class _Ctx:
            def __enter__(self_):
                global RECORDER
                torch.cuda.synchronize()
                seg._stream = torch.cuda.Stream()
                seg._stream.wait_stream(torch.cuda.current_stream())
                self_.ctx = torch.cuda.stream(seg._stream)
                self_.ctx.__enter__()
                RECORDER = seg
                seg._begin()
                return seg

            def __exit__(self_, *exc):
                global RECORDER
                RECORDER = None
                if seg._cur is not None:
                    if exc[0] is None:
                        seg._end()
                    else:
                        try:
                            seg._cur.capture_end()
                        except Exception:
                            pass
                self_.ctx.__exit__(*exc)
                torch.cuda.current_stream().wait_stream(seg._stream)
                return False

        return _Ctx()

    def eager(self, fn):
        """Called while recording: close the current graph segment, note `fn` as the eager call that runs
        between the two segments at every replay, open the next segment.  `fn` is NOT run while recording:
        nothing is -- the captured kernels around it only get recorded, so its inputs hold no data yet --
        and a recording pass without collectives cannot fall out of step with its peers if the capture
        fails on one rank (the fallback decision is taken collectively afterwards)."""
        self._end()
        self.items.append(("call", fn))
        self._begin()

    def replay(self):
        for kind, obj in self.items:
            if kind == "graph":
                obj.replay()
            else:
                obj()


RECORDER = None                     # the SegmentedGraph being recorded, if any

# What this rank put on the wire since reset_comm(): collective -> [calls, bytes this rank contributed].  Counted where the
# collective is ISSUED (eagerly, also between replayed graph segments), so bench.py --gpus N can print bytes per collective
# next to the timing and a SCALE record checks itself.
COMM = {}


def _note(name, t):
    c = COMM.setdefault(name, [0, 0])
    c[0] += 1
    c[1] += t.numel() * t.element_size()


def reset_comm():
    COMM.clear()


def comm_stats(steps=1):
    return {k: {"calls_per_step": v[0] / steps, "bytes_per_rank_per_call": v[1] // max(v[0], 1),
                "bytes_per_rank_per_step": v[1] / steps} for k, v in sorted(COMM.items())}


def _collective(call):
    if RECORDER is not None:
        RECORDER.eager(call)        # graph break: the collective stays an eager call between two segments
    else:
        call()


# Collectives that do not block the compute stream.  PF_SHARD_ASYNC=0 restores the blocking calls (A/B).
# `issue` posts the collective with async_op=True: the backend runs it on its own communication stream behind what the compute stream
# has enqueued so far; the compute stream goes on with kernels that do not need the result, and `wait()` -- placed right in front of the
# consumer -- makes it wait for the collective.  Every rank issues the collectives of a step in the SAME order (the order of the
# `issue` calls, which is program order on every rank); where a rank waits is its own business.  Under SegmentedGraph both the issue and the
# wait are eager calls between graph segments.
ASYNC = os.environ.get("PF_SHARD_ASYNC", "1") != "0"


class Pending:
    """Handle of a posted collective: wait() before the first kernel that reads its output (idempotent)."""

    def __init__(self, issue):
        self._work = None
        self._waited = False

        def post():
            self._work = issue()
        _collective(post)

    def wait(self):
        if self._waited:
            return
        self._waited = True

        def block():
            w, self._work = self._work, None
            if w is not None:
                w.wait()            # NCCL: the current stream waits for the communication stream; gloo: the host blocks
        _collective(block)


class _Ready:
    """A result that needed no collective (one rank per CFG half), with the interface of a posted one."""

    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


class _Gathered:
    def __init__(self, pending, out, counts, rows, P, G):
        self.pending, self.out, self.counts, self.rows, self.P, self.G = pending, out, counts, rows, P, G

    def result(self):
        """[m * P, C] in view order; waits for the all-gather first."""
        self.pending.wait()
        if len(set(self.counts)) == 1:
            return self.out
        return torch.cat([self.out[g * self.rows:g * self.rows + self.counts[g] * self.P] for g in range(self.G)])

    def finish(self):
        """A rank that does not read the gathered tokens still has to complete the collective before its buffers go away."""
        self.pending.wait()


def gather_view_tokens_async(x_local, shard, P=None, C=None, dtype=None, device=None):
    """gather_view_tokens posted without blocking the compute stream: returns a handle whose result() waits.  A rank without views
    may post it EARLY (x_local = None: it contributes an empty block and needs P, C, dtype, device) -- the panorama-only owner posts the
    all-gather of an EPA block before it runs the panorama resnets of that level and collects the tokens when it gets to the block."""
    if shard.G == 1:
        return _Ready(x_local)
    counts = shard.counts
    if x_local is None:
        x_local = torch.empty(0, C, dtype=dtype, device=device)
    x_local = x_local.contiguous()
    if P is None:
        P = x_local.shape[0] // counts[shard.g]
    rows = shard.vmax * P
    if x_local.shape[0] != rows:
        padded = torch.empty(rows, x_local.shape[1], dtype=x_local.dtype, device=x_local.device)
        padded[:x_local.shape[0]] = x_local
        x_local = padded
    out = torch.empty(shard.G * rows, x_local.shape[1], dtype=x_local.dtype, device=x_local.device)

    def issue():
        _note("all_gather view tokens (EPA, group of %d)" % shard.G, x_local)
        return dist.all_gather_into_tensor(out, x_local, group=shard.group, async_op=True)
    return _Gathered(Pending(issue), out, counts, rows, P, shard.G)


class _Shared:
    def __init__(self, pending, buf):
        self.pending, self.buf = pending, buf

    def result(self):
        if self.pending is not None:
            self.pending.wait()
        return self.buf


def share_pano_tokens_async(x, rows, cols, like, shard):
    """share_pano_tokens posted without blocking the compute stream (result() waits)."""
    if shard.pano_g is None or shard.G == 1:
        return _Shared(None, x)
    buf = x.contiguous() if x is not None else torch.empty(rows, cols, dtype=like.dtype, device=like.device)

    def issue():
        _note("broadcast panorama tokens (EPA, group of %d)" % shard.G, buf)
        return dist.broadcast(buf, src=shard.pano_src, group=shard.group, async_op=True)
    return _Shared(Pending(issue), buf)


def gather_view_tokens(x_local, shard, P=None):
    """All-gather [views_local * P, C] token blocks of the CFG half in view order -> [m * P, C].
    Unequal view counts (panorama-rank layout) travel padded to the largest group and are compacted.
    P (tokens per view) must be given by a rank without views (x_local is then an empty [0, C] tensor)."""
    if shard.G == 1:
        return x_local
    x_local = x_local.contiguous()
    counts = shard.counts
    if P is None:
        P = x_local.shape[0] // counts[shard.g]
    rows = shard.vmax * P
    if x_local.shape[0] != rows:
        padded = torch.empty(rows, x_local.shape[1], dtype=x_local.dtype, device=x_local.device)
        padded[:x_local.shape[0]] = x_local
        x_local = padded
    out = torch.empty(shard.G * rows, x_local.shape[1], dtype=x_local.dtype, device=x_local.device)
    def call():
        _note("all_gather view tokens (EPA, group of %d)" % shard.G, x_local)
        dist.all_gather_into_tensor(out, x_local, group=shard.group)
    _collective(call)
    if len(set(counts)) == 1:
        return out
    return torch.cat([out[g * rows:g * rows + counts[g] * P] for g in range(shard.G)])


def share_pano_tokens(x, rows, cols, like, shard):
    """The panorama owner's [rows, cols] tensor on every rank of the CFG half (x is None elsewhere)."""
    if shard.pano_g is None or shard.G == 1:
        return x
    buf = x.contiguous() if x is not None else torch.empty(rows, cols, dtype=like.dtype, device=like.device)
    def call():
        _note("broadcast panorama tokens (EPA, group of %d)" % shard.G, buf)
        dist.broadcast(buf, src=shard.pano_src, group=shard.group)
    _collective(call)
    return buf


# ---- query-split of the panorama branch's big self-attentions ("mandatory for cfg 4") -----------------------------
# At configs[3] (128 x 256 panorama latent) the panorama owner of a CFG half is the slowest rank: its two + three level-0
# self-attentions walk 32 768 keys for 32 768 queries (11 ms of its 28 ms) while the view ranks of the half
# finish earlier.  With this on (panorama-rank layout, tokens >= PF_SHARD_ATTN_MIN_TOKENS) the owner broadcasts the layer-normed
# TOKENS of such an attention inside the half, EVERY rank of the half projects q | k | V^T and computes the rows [g nq / G, (g + 1) nq / G)
# of the output, one all-gather returns them.  Per attention and rank: C nq 16-bit words in (21 MB at C = 320, nq = 32 768; the first
# version sent q | k and V^T: 63 MB), C nq / G words out; the view ranks call help_pano_attention right after their own self-attention
# of the same UNet position, so every rank issues the same collective sequence.  A query row's result does not depend on which rank
# computes it: replicas stay bit-identical.
ATTN_SPLIT_MIN_TOKENS = int(os.environ.get("PF_SHARD_ATTN_MIN_TOKENS", "16384"))
# ... and from this many ranks per CFG half: with G = 2 the one view rank of the half (13-20 views) is the slowest already, and
# handing it half of the panorama's attention makes the step slower (cfg 4, 4 ranks, one-GPU simulation: 34.1 -> 41.5 ms; 8 ranks:
# owner 33.0 -> 25.7, view ranks 18.0 -> 21.8: profiles/r5e_sim_cfg4_attn_split.txt)
ATTN_SPLIT_MIN_GROUP = int(os.environ.get("PF_SHARD_ATTN_MIN_GROUP", "3"))


def splits_pano_attention(shard, tokens):
    """The rule both sides evaluate (owner: in its panorama self-attention; view ranks: after their own)."""
    return shard is not None and shard.pano_g is not None and _split_rule(shard.G, tokens)


def split_pano_attention(shard, a, tokens, nq, dtype=None, device=None):
    """ONE panorama self-attention of the owner, query-split over its CFG half.  a: the packed attention of the PANORAMA UNet at
    this position (every rank packs that UNet: the weights are replicated anyway); tokens [nq, C]: its layer-normed input on the
    owner, None elsewhere.  The tokens are broadcast (C nq 16-bit words: 21 MB at C = 320, nq = 32 768 -- a third of what q | k and
    V^T would be), every rank projects q | k | V^T itself (the same launch on the same data: bit-identical), computes the rows
    [g nq / G, (g + 1) nq / G) of the output, and one all-gather returns [nq, C] to every rank (only the owner uses it)."""
    from . import engine
    G, g = shard.G, shard.g
    C = a.dim
    if tokens is None:
        tokens = torch.empty(nq, C, dtype=dtype, device=device)
    tokens = tokens.contiguous()

    def bcast():
        _note("broadcast panorama self-attention tokens (query split, group of %d)" % G, tokens)
        dist.broadcast(tokens, src=shard.pano_src, group=shard.group)
    _collective(bcast)
    qk, vt = engine.self_qkv(a, tokens, 1, nq)
    rows = nq // G
    r0 = g * rows
    ld = 2 * C
    o_loc = ops.attention(qk[r0:r0 + rows], qk[:, C:], vt, 1, a.heads, C // a.heads, rows, nq, q_ld=ld, k_ld=ld, vt_ld=vt.shape[-1],
                          q_bs=rows * ld, k_bs=nq * ld, vt_bs=vt.shape[1] * vt.shape[2])
    o_loc = o_loc.reshape(rows, C)
    out = torch.empty(nq, C, dtype=o_loc.dtype, device=o_loc.device)

    def gather():
        _note("all_gather panorama attention rows (query split, group of %d)" % G, o_loc)
        dist.all_gather_into_tensor(out, o_loc, group=shard.group)
    _collective(gather)
    return out


def help_pano_attention(shard, pano_t_pack, tokens, like):
    """A view rank's share of the owner's panorama self-attention; pano_t_pack: the packed transformer of the PANORAMA UNet at
    the UNet position of the view rank's own self-attention."""
    split_pano_attention(shard, pano_t_pack.attn1, None, tokens, dtype=pano_t_pack.dtype, device=like.device)


def gather_eps(eps_local, pano_eps_local, shard, pano_shape=None):
    """-> eps [2, m, ...] and pano_eps [2, 1, ...] on every rank (rank order == (cfg, view group)).
    pano_eps_local is None on ranks without the panorama branch (pano_shape = its shape there)."""
    e = eps_local.contiguous()                            # [1, views_local, 4, h, w]
    counts = shard.counts
    if e.shape[1] != shard.vmax:                          # unequal groups travel padded to the largest
        pad = torch.zeros(1, shard.vmax, *e.shape[2:], dtype=e.dtype, device=e.device)
        pad[:, :e.shape[1]] = e
        e = pad
    out = torch.empty(shard.world, *e.shape[1:], dtype=e.dtype, device=e.device)
    _note("all_gather eps views (world)", e)
    dist.all_gather_into_tensor(out, e)                   # concatenated along dim 0 in rank order
    if len(set(counts)) == 1:
        eps = out.view(2, shard.m, *e.shape[2:])          # [(c, g), m/G, ...] is [2, m, ...] in memory
    else:
        eps = torch.stack([torch.cat([out[c * shard.G + g, :counts[g]] for g in range(shard.G)]) for c in range(2)])
    if pano_eps_local is None:
        p = torch.zeros(*pano_shape, dtype=e.dtype, device=e.device)
    else:
        p = pano_eps_local.contiguous()                   # [1, 1, 4, H, W]
    pout = torch.empty(shard.world * p.shape[0], *p.shape[1:], dtype=p.dtype, device=p.device)
    _note("all_gather eps panorama (world)", p)
    dist.all_gather_into_tensor(pout, p)
    pano_eps = pout[(shard.pano_g or 0)::shard.G].contiguous()    # the owner's (replicas are identical otherwise)
    return eps, pano_eps


class ShardedDenoiseLoop(DenoiseLoop):
    """DenoiseLoop whose denoiser call computes only this rank's (CFG sample, view group)."""

    def __init__(self, model, shard, *a, **k):
        super().__init__(model, *a, **k)
        self.shard = shard
        self.layout_desc = "cfg2 x viewgroups%d %s" % (shard.G, "views " + "/".join(map(str, shard.counts)) +
                                                       (" (group 0 owns the panorama)" if shard.pano_g is not None else ""))
        model.shard = shard

    def _local(self, cams):
        s = self.shard
        v0, v1 = s.views
        c = s.cfg
        ts = self.tstep[:1, v0:v1] if v1 > v0 else self.tstep[:1, :1]     # a panorama-only owner still needs t
        lay = self._layout_for(cams)                 # panorama ControlNet condition (CFG pair): this rank's sample,
        lay = lay[c:c + 1] if (lay is not None and s.has_pano) else None   # on the ranks that run the panorama branch
        return self.model(self.lat[:, v0:v1].contiguous(), self.pano, ts,
                          self.prompt[c:c + 1, v0:v1], self.pano_prompt[c:c + 1], cams, None, lay)

    def _gather(self, out):
        return gather_eps(out[0], out[1], self.shard, pano_shape=(1,) + tuple(self.pano.shape[1:]))

    def _denoise(self, cams):
        return self._gather(self._local(cams))

    def _denoise_graphed(self, cams):
        """This rank's share of the denoiser as hipGraph segments between the EPA collectives (one
        SegmentedGraph per rotation offset); the epsilon all-gather follows eagerly."""
        key = tuple(float(v) for v in cams["theta"].reshape(-1))
        g = self.graphs.get(key)
        if g is None:
            from .engine import EPATables
            self._local(cams)                        # warm-up: tables, kernel attributes, communicator
            torch.cuda.synchronize()
            seg, out, ok = SegmentedGraph(), None, 1
            try:
                with EPATables.pinned(), seg.record():   # no collective runs in here (SegmentedGraph.eager)
                    out = self._local(cams)
            except RuntimeError as e:                # capture refused (driver / communicator state)
                import sys
                print("panfusion_amd.sharding: hipGraph capture failed on rank %d (%s)" % (self.shard.rank, e), file=sys.stderr)
                torch.cuda.synchronize()
                ok = 0
            # the fallback is decided by ALL ranks together: if the capture failed anywhere, everybody drops
            # its graphs and launches eagerly from here on (same kernels, same collective sequence on every rank)
            flag = torch.tensor([ok], dtype=torch.int32, device=self.pano.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                self.use_graphs = False
                self.graphs.clear()
                return self._denoise(cams)
            g = (seg, out, self._graph_keepalive())
            self.graphs[key] = g
        g[0].replay()
        return self._gather(g[1])


def build_sharded(build_model, build_inputs, dev, dtype, cfg, m, lat_hw, pano_hw, cams_deg, steps, use_graphs,
                  precision=None, layout_cond=False, layout=None):
    """layout_cond / layout: BASELINE.json configs[4] -- the model carries the panorama ControlNet and the loop its condition
    image (MVGenModel.py:75-83, PanFusion.py:150-153); the ranks that run the panorama branch run the ControlNet on their CFG
    sample's copy of the image (ShardedDenoiseLoop._local)."""
    if layout_cond and layout is None:
        raise ValueError("a layout-conditioned model needs its condition image")
    shard = make_shard(m, pano_hw=pano_hw, lat_hw=lat_hw, layout_cond=layout_cond)
    model = build_model(dev, dtype, cfg, layout_cond=layout_cond, precision=precision)
    inputs = build_inputs(dev, m, lat_hw, pano_hw, cfg["cross_attention_dim"], cams_deg)
    loop = ShardedDenoiseLoop(model, shard, *inputs, steps=steps, use_graphs=use_graphs, pano_layout_cond=layout)
    return model, loop
