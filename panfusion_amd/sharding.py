"""Multi-GPU sharding of one denoising step (one process per GPU, ``torch.distributed`` over
RCCL/xGMI on the GPU box, gloo in the CPU tests).

The reference has no intra-step parallelism (views are a batch dimension on one GPU, SURVEY.md
§2.1); this is new design.  Independent units of a step:
  * the two classifier-free-guidance samples (independent everywhere, EPA included);
  * the m perspective views (independent everywhere except inside EPA).
Layout for N ranks (N even): 2 CFG halves x G = N/2 view groups; rank r -> (c, g) = (r // G, r % G),
views [g*m/G, (g+1)*m/G).  N = 2 is the pure CFG split: no traffic inside the step.

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

    @property
    def views(self):
        per = self.m // self.G
        return self.g * per, (self.g + 1) * per


def plan(world, rank, m):
    """(cfg, g, G) layout of `rank`; world must be even (CFG pair) and m divisible by G."""
    if world < 2 or world % 2:
        raise ValueError("sharded step needs an even number of ranks (CFG pair x view groups), got %d" % world)
    G = world // 2
    if m % G:
        raise ValueError("%d views do not divide over %d view groups" % (m, G))
    return ShardInfo(rank=rank, world=world, cfg=rank // G, g=rank % G, G=G, m=m)


def make_shard(m):
    """Build this process' ShardInfo and the process group of its CFG half (collective call:
    every rank creates every group, in the same order)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    info = plan(world, rank, m)
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
        """Called while recording: close the current graph segment, run `fn` eagerly (it is re-run at
        every replay), open the next segment."""
        self._end()
        fn()
        self.items.append(("call", fn))
        self._begin()

    def replay(self):
        for kind, obj in self.items:
            if kind == "graph":
                obj.replay()
            else:
                obj()


RECORDER = None                     # the SegmentedGraph being recorded, if any


def gather_view_tokens(x_local, shard):
    """All-gather [rows_local, C] token blocks of the CFG half in view order -> [G * rows_local, C]."""
    if shard.G == 1:
        return x_local
    x_local = x_local.contiguous()
    out = torch.empty(shard.G * x_local.shape[0], x_local.shape[1], dtype=x_local.dtype, device=x_local.device)
    call = lambda: dist.all_gather_into_tensor(out, x_local, group=shard.group)
    if RECORDER is not None:
        RECORDER.eager(call)        # graph break: the collective stays an eager call between two segments
    else:
        call()
    return out


def gather_eps(eps_local, pano_eps_local, shard):
    """-> eps [2, m, ...] and pano_eps [2, 1, ...] on every rank (rank order == (cfg, view group))."""
    e = eps_local.contiguous()                            # [1, m/G, 4, h, w]
    out = torch.empty(shard.world * e.shape[0], *e.shape[1:], dtype=e.dtype, device=e.device)
    dist.all_gather_into_tensor(out, e)                   # concatenated along dim 0 in rank order
    eps = out.view(2, shard.m, *e.shape[2:])              # [(c, g), m/G, ...] is [2, m, ...] in memory
    p = pano_eps_local.contiguous()                       # [1, 1, 4, H, W]
    pout = torch.empty(shard.world * p.shape[0], *p.shape[1:], dtype=p.dtype, device=p.device)
    dist.all_gather_into_tensor(pout, p)
    pano_eps = pout[::shard.G].contiguous()               # replicas inside a CFG half are identical
    return eps, pano_eps


class ShardedDenoiseLoop(DenoiseLoop):
    """DenoiseLoop whose denoiser call computes only this rank's (CFG sample, view group)."""

    def __init__(self, model, shard, *a, **k):
        super().__init__(model, *a, **k)
        self.shard = shard
        self.layout = "cfg2 x viewgroups%d" % shard.G
        model.shard = shard

    def _local(self, cams):
        s = self.shard
        v0, v1 = s.views
        c = s.cfg
        return self.model(self.lat[:, v0:v1].contiguous(), self.pano, self.tstep[:1, v0:v1],
                          self.prompt[c:c + 1, v0:v1], self.pano_prompt[c:c + 1], cams)

    def _denoise(self, cams):
        return gather_eps(*self._local(cams), self.shard)

    def _denoise_graphed(self, cams):
        """This rank's share of the denoiser as hipGraph segments between the EPA collectives (one
        SegmentedGraph per rotation offset); the epsilon all-gather follows eagerly."""
        key = tuple(float(v) for v in cams["theta"].reshape(-1))
        g = self.graphs.get(key)
        if g is None:
            self._local(cams)                        # warm-up: tables, kernel attributes, communicator
            torch.cuda.synchronize()
            seg = SegmentedGraph()
            try:
                with seg.record():
                    out = self._local(cams)
            except RuntimeError as e:                # capture refused (driver / communicator state): the same
                import sys                           # kernels are launched eagerly instead -- slower host side, same results
                print("panfusion_amd.sharding: hipGraph capture failed (%s); launching eagerly" % e, file=sys.stderr)
                torch.cuda.synchronize()
                self.use_graphs = False
                return self._denoise(cams)
            g = (seg, out)
            self.graphs[key] = g
        g[0].replay()
        return gather_eps(*g[1], self.shard)


def build_sharded(build_model, build_inputs, dev, dtype, cfg, m, lat_hw, pano_hw, cams_deg, steps, use_graphs):
    shard = make_shard(m)
    model = build_model(dev, dtype, cfg)
    inputs = build_inputs(dev, m, lat_hw, pano_hw, cfg["cross_attention_dim"], cams_deg)
    loop = ShardedDenoiseLoop(model, shard, *inputs, steps=steps, use_graphs=use_graphs)
    return model, loop
