"""Multi-rank sharding of the denoising step over torch.distributed (gloo, CPU) with the test
double tests/fake_ops.py standing in for the HIP kernels: layout, collectives and loop logic of
panfusion_amd/sharding.py must reproduce the single-process result."""
import importlib
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_loop(sharded, steps=2, split=None, layout=None, precision="fast", controlnet=False, attn_min=None):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import fake_ops
    from conftest import build_tiny_oracle, cam4, golden
    from test_engine_logic_cpu import MODS, hip_model
    for name in MODS + ["panfusion_amd.sharding"]:
        importlib.import_module(name).ops = fake_ops
    from panfusion_amd import sharding
    from panfusion_amd.pipeline import DenoiseLoop
    if attn_min is not None:                     # query-split of the panorama self-attentions from this many tokens (toy sizes),
        sharding.ATTN_SPLIT_MIN_TOKENS = attn_min    # also with two ranks per CFG half (the default starts at three)
        sharding.ATTN_SPLIT_MIN_GROUP = 2
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    om = build_tiny_oracle()
    kw = {}
    if controlnet:      # BASELINE.json configs[4]: a panorama ControlNet + its layout image, rolled with the panorama every step
        from oracle import sd2_unet as U
        from panfusion_amd.models.pano import MultiViewBaseModel
        cn = U.ControlNetModel.from_unet(om.pano_unet)
        U.init_synthetic(cn.controlnet_cond_embedding, 71)
        U.init_synthetic(cn.controlnet_down_blocks, 72)
        U.init_synthetic(cn.controlnet_mid_block, 73)
        model = MultiViewBaseModel(om.unet, om.pano_unet, None, cn, True, compute_dtype=torch.float32, precision=precision)
        model.load_state_dict({k: v for k, v in om.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
        kw["pano_layout_cond"] = torch.rand(1, 1, 3, 128, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1
    else:
        model = hip_model(om, precision=precision)
    cam1 = {k: v[None] for k, v in cam4().items()}
    args = (t("latents")[:1], t("pano_latent")[:1], t("prompt_embd"), t("pano_prompt_embd"), cam1)
    if sharded:
        loop = sharding.ShardedDenoiseLoop(model, sharding.make_shard(4, layout=layout, split=split), *args, steps=steps, **kw)
    else:
        loop = DenoiseLoop(model, *args, steps=steps, **kw)
    return loop.run()


def _worker(rank, world, port, out, split=None, layout=None, precision="fast", controlnet=False, attn_min=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lat, pano = _run_loop(True, split=split, layout=layout, precision=precision, controlnet=controlnet, attn_min=attn_min)
        torch.save((lat, pano), os.path.join(out, "r%d.pt" % rank))
        from panfusion_amd import sharding
        torch.save(sharding.comm_stats(2), os.path.join(out, "comm%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_plan_layout():
    from panfusion_amd import sharding
    got = [(s.cfg, s.g, s.G) for s in (sharding.plan(2, r, 20) for r in range(2))]
    assert got == [(0, 0, 1), (1, 0, 1)]
    got = [(s.cfg, s.g, s.G, s.views) for s in (sharding.plan(4, r, 20, layout="even") for r in range(4))]
    assert got == [(0, 0, 2, (0, 10)), (0, 1, 2, (10, 20)), (1, 0, 2, (0, 10)), (1, 1, 2, (10, 20))]
    s = sharding.plan(8, 6, 20, layout="even")
    assert (s.cfg, s.g, s.G, s.views) == (1, 2, 4, (10, 15))
    with pytest.raises(ValueError):
        sharding.plan(3, 0, 20)
    with pytest.raises(ValueError):
        sharding.plan(16, 0, 20, layout="even")          # 20 views do not split into 8 groups
    # panorama-rank layout: group 0 of a CFG half owns the panorama branch and fewer views; the split
    # minimises the slowest group under the measured time model (panorama branch = 6.2 views of time at cfg 2, round-6 fit at the final build)
    assert sharding.pano_rank_split(20, 2) == (7, 13)
    assert sharding.pano_rank_split(20, 4) == (0, 7, 7, 6)      # a panorama-only owner is cheaper than one with a view
    assert sharding.pano_rank_split(20, 1) is None
    s = sharding.plan(8, 5, 20)                           # "auto" picks it whenever it is faster than replicating
    assert (s.cfg, s.g, s.counts, s.views, s.has_pano, s.pano_src, s.vmax) == (1, 1, (0, 7, 7, 6), (0, 7), False, 4, 7)
    s = sharding.plan(8, 4, 20)
    assert s.has_pano and s.views == (0, 0) and not s.has_views
    s = sharding.plan(4, 1, 20)
    assert s.pano_g == 0 and s.views == (7, 20) and not s.has_pano
    assert sharding.plan(4, 1, 20, layout="even").pano_g is None and sharding.plan(8, 1, 20, layout="even").views == (5, 10)
    s = sharding.plan(8, 0, 20, split=(0, 7, 7, 6))      # explicit split only: a panorama-only owner
    assert s.has_pano and not s.has_views and s.views == (0, 0) and sharding.plan(8, 1, 20, split=(0, 7, 7, 6)).views == (0, 7)
    with pytest.raises(ValueError):
        sharding.plan(8, 0, 20, split=(7, 0, 7, 6))      # only the owner may go without views
    s = sharding.plan(16, 3, 20)                          # 8 groups: 0 / 3 3 3 3 3 3 2
    assert sum(s.counts) == 20 and s.counts[0] == 0 and max(s.counts[1:]) - min(s.counts[1:]) <= 1


def test_plan_uses_the_time_model_of_the_configuration():
    """cfg 4 (128 x 256 panorama latent): the panorama branch costs three times that of cfg 2 in time (5 x for an owner without
    views) -- the split follows the configuration."""
    from panfusion_amd import sharding
    tm2, tm4 = sharding.time_model((64, 128), (64, 64)), sharding.time_model((128, 256), (64, 64))
    assert tm2["measured"] and tm4["measured"] and tm4["pano"] > 3 * tm2["pano"] and tm4["pano_only"] > 4 * tm2["pano_only"]
    assert not sharding.time_model((32, 64), (32, 32))["measured"]
    s2, s4 = sharding.plan(4, 0, 20, pano_hw=(64, 128), lat_hw=(64, 64)), sharding.plan(4, 0, 20, pano_hw=(128, 256), lat_hw=(64, 64))
    assert s2.counts == (7, 13) and s4.counts == (0, 20)      # at cfg 4 the owner of 4 ranks keeps no views at all
    assert sharding.time_model((64, 128), (64, 64), True)["pano_only"] > 2 * tm2["pano_only"]       # cfg 5: the ControlNet rides on the owner
    assert sharding.plan(8, 0, 20, pano_hw=(128, 256), lat_hw=(64, 64)).counts == (0, 7, 7, 6)
    # the query split of the 32 768-token self-attentions (G >= 3): the owner sheds attn (1 - 1 / G), every other rank takes on attn_help / G
    # -- two constants since round 6 (measured 3.13 / 2.94 ms at G = 4: profiles/r6_time_model.json)
    assert abs(sharding.step_time_ms((0, 7, 7, 6), False, tm4) - (tm4["base"] + tm4["pano_only"] - 0.75 * tm4["attn"])) < 1e-6
    assert abs(sharding.helper_cost(7, tm4, 4) * tm4["per_view"] - (7 * tm4["per_view"] + tm4["attn_help"] / 4)) < 1e-6
    # G = 2: no split (PF_SHARD_ATTN_MIN_GROUP = 3); the 20-view rank is the slowest (20 view units > pano_only / per_view = 16.6 of the owner alone)
    assert abs(sharding.split_cost((0, 20), False, tm4) - 20.0) < 1e-9 and sharding.owner_cost(0, tm4, 2) * tm4["per_view"] == pytest.approx(tm4["pano_only"])
    assert sharding.step_time_ms((0, 7, 7, 6), False, tm4) > 1.35 * sharding.step_time_ms((0, 7, 7, 6), False, tm2)
    # the model reproduces what tools/fit_time_model.py timed at HEAD (simulated ranks, no wire time)
    assert abs(sharding.step_time_ms((0, 7, 7, 6), False, tm2) - 13.72) < 0.05 and abs(sharding.step_time_ms((0, 7, 7, 6), False, tm4) - 19.69) < 0.05


def test_plan_minimises_the_modelled_slowest_rank():
    """whatever constants TIME_MODEL carries, the split plan() returns is the arg-min of the modelled slowest rank over
    EVERY panorama-rank split (brute force over the owner's view count, the rest spread evenly or not), and "auto" takes the even layout
    only where the model says it is at least as fast."""
    import itertools
    from panfusion_amd import sharding
    for key in sharding.TIME_MODEL:
        tm = sharding.time_model(*key)
        for world in (4, 6, 8, 10):
            G = world // 2
            s = sharding.plan(world, 0, 20, pano_hw=key[0], lat_hw=key[1], layout_cond=key[2])
            got = sharding.split_cost(s.counts, s.pano_g is None, tm)
            best = None
            for m0 in range(0, 20 - (G - 1) + 1):
                rest = 20 - m0
                q, r = divmod(rest, G - 1)
                cand = (m0,) + tuple(q + (1 if i < r else 0) for i in range(G - 1))
                c = sharding.split_cost(cand, False, tm)
                best = c if best is None else min(best, c)
            if G <= 3:                                   # small enough to try every composition of 20 into G parts with >= 1 view off the owner
                for rest in itertools.product(range(1, 20), repeat=G - 1):
                    if sum(rest) <= 20:
                        best = min(best, sharding.split_cost((20 - sum(rest),) + rest, False, tm))
            if 20 % G == 0:
                best = min(best, sharding.split_cost((20 // G,) * G, True, tm))
            assert got <= best + 1e-9, (key, world, s.counts, got, best)


def test_planner_and_runtime_share_the_split_rule():
    """with G = 3 or 5 ranks per CFG half 32 768 tokens are not a multiple of 32 G -- the run time does not split the panorama
    self-attentions, so the planner must not book the saving."""
    from panfusion_amd import sharding
    tm4 = sharding.time_model((128, 256), (64, 64))
    for G in (2, 3, 4, 5, 8):
        shard = sharding.ShardInfo(rank=0, world=2 * G, cfg=0, g=0, G=G, m=20, pano_g=0, split=(0,) + (1,) * (G - 1))
        runtime = sharding.splits_pano_attention(shard, 128 * 256)
        shed, extra = sharding._attn_shares(tm4, G)
        assert (shed > 0) == runtime and (extra > 0) == runtime, (G, runtime, shed, extra)
    assert not sharding.splits_pano_attention(sharding.ShardInfo(rank=0, world=6, cfg=0, g=0, G=3, m=20, pano_g=0), 128 * 256)


def test_sharded_loop_with_panorama_controlnet_equals_single_process():
    """BASELINE.json configs[4] is an 8-GPU configuration: the sharded loop must run the panorama ControlNet on the ranks that own
    the panorama branch, with the layout image rolled as the panorama is (MVGenModel.py:75-83, PanFusion.py:150-153)."""
    base = _run_loop(False)
    want = _run_loop(False, controlnet=True)
    assert float((want[1] - base[1]).norm() / base[1].norm()) > 1e-3, "the ControlNet must change the result for the test to mean anything"
    for world, split in ((4, None), (4, (1, 3))):
        with tempfile.TemporaryDirectory() as out:
            mp.spawn(_worker, args=(world, _free_port(), out, split, None, "fast", True), nprocs=world, join=True)
            res = [torch.load(os.path.join(out, "r%d.pt" % r)) for r in range(world)]
        rel = lambda a, b: float((a - b).norm() / b.norm())
        for lat, pano in res:
            assert rel(lat, want[0]) < 1e-4 and rel(pano, want[1]) < 1e-4, (world, split, rel(lat, want[0]), rel(pano, want[1]))
        assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])


@pytest.mark.parametrize("world,split,layout,precision", [
    (2, None, None, "fast"), (4, None, "even", "fast"), (4, None, None, "fast"), (6, (2, 1, 1), None, "fast"),
    (6, (0, 2, 2), None, "fast"), (6, None, None, "fast"), (8, None, None, "fast"),
    (4, None, "even", "mixed"), (6, (0, 2, 2), None, "mixed")])       # fp32 streams + split-precision GEMMs, sharded
def test_sharded_loop_equals_single_process(world, split, layout, precision):
    """(4, auto) = (0, 4), (6, auto) = (0, 2, 2), (8, auto) = (0, 2, 1, 1) and (6, (2, 1, 1)): the panorama-rank layout -- group 0 of a CFG half owns the panorama branch and
    fewer views, the other ranks run the view branch only and receive the panorama tokens by broadcast (unequal,
    padded gathers).  (4, "even"): views split evenly, panorama branch replicated.  (6, (0, 2, 2)): a panorama-only
    owner -- no view branch on ranks 0 / 3, they contribute empty blocks to the gathers."""
    want = _run_loop(False, precision=precision)
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(world, _free_port(), out, split, layout, precision), nprocs=world, join=True)
        res = [torch.load(os.path.join(out, "r%d.pt" % r)) for r in range(world)]
        comm = torch.load(os.path.join(out, "comm0.pt"))
    # what bench.py --gpus N prints per collective (sharding.comm_stats): 7 EPA blocks per step, each one all-gather of the
    # view tokens inside the CFG half (+ one broadcast of the panorama tokens in the panorama-rank layout), 2 epsilon all-gathers
    if world > 2:
        G = world // 2
        assert comm["all_gather view tokens (EPA, group of %d)" % G]["calls_per_step"] == 7
        assert ("broadcast panorama tokens (EPA, group of %d)" % G in comm) == (layout != "even")
    else:
        assert not any("EPA" in k for k in comm)          # the pure CFG split: no traffic inside the step
    assert comm["all_gather eps views (world)"]["calls_per_step"] == 1 and comm["all_gather eps panorama (world)"]["calls_per_step"] == 1
    assert comm["all_gather eps panorama (world)"]["bytes_per_rank_per_call"] == want[1].numel() * 4
    for lat, pano in res:                       # every rank holds the full, identical latents
        rel = lambda a, b: float((a - b).norm() / b.norm())       # fp32 round-off of differently batched convs
        assert rel(lat, want[0]) < 1e-4 and rel(pano, want[1]) < 1e-4, (rel(lat, want[0]), rel(pano, want[1]))
    # replicas must not drift apart: every rank applies the same update to the same gathered epsilons
    assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])


@pytest.mark.parametrize("world,split,precision", [(4, None, "fast"), (4, (1, 3), "mixed"), (8, None, "fast")])
def test_panorama_self_attention_split_over_the_cfg_half(world, split, precision):
    """SURVEY.md 8e, configs[3]: the panorama owner's big self-attentions are query-split over the ranks of its CFG half
    (sharding.split_pano_attention: broadcast of the layer-normed TOKENS, every rank projects q | k | V^T itself and computes nq / G rows
    of the output, one all-gather returns them) -- here forced at toy
    sizes (>= 128 tokens: the 16 x 32 and 8 x 16 levels).  Same result as the single process, bit-identical replicas, and the
    collectives show up in the accounting bench.py prints.  CPU / gloo coverage only; the graph-segment side (SegmentedGraph with a split
    attention on an owner that also holds views and runs the panorama branch on its side stream) is exercised by the real-kernel dry run
    profiles/r6_dist_dry_cfg4_owner_with_views_split.txt (tools/gpu_dist_dry.sh, graphs_in_use True on every rank)."""
    want = _run_loop(False, precision=precision)
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(world, _free_port(), out, split, None, precision, False, 128), nprocs=world, join=True)
        res = [torch.load(os.path.join(out, "r%d.pt" % r)) for r in range(world)]
        comm = [torch.load(os.path.join(out, "comm%d.pt" % r)) for r in range(world)]
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for lat, pano in res:
        assert rel(lat, want[0]) < 1e-4 and rel(pano, want[1]) < 1e-4, (rel(lat, want[0]), rel(pano, want[1]))
    assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
    G = world // 2
    key = "all_gather panorama attention rows (query split, group of %d)" % G
    n_split = comm[0][key]["calls_per_step"]
    assert n_split >= 5                                     # the five level-0 self-attentions at least (512 tokens at 16 x 32)
    for c in comm:                                          # owner and helpers issue the same sequence
        assert c[key]["calls_per_step"] == n_split
        assert c["broadcast panorama self-attention tokens (query split, group of %d)" % G]["calls_per_step"] == n_split


def _batch2_worker(rank, world, port, out):
    """The denoiser forward of a sharded rank on a TWO-prompt batch (b = 2 samples of the rank's CFG half)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, args = _batch2_inputs()
        from panfusion_amd import sharding
        shard = sharding.make_shard(4, split=(1, 3))
        model.shard = shard
        v0, v1 = shard.views
        lat, pano, ts, pe, ppe, cams = args
        s, ps = model(lat[:, v0:v1].contiguous(), pano, ts[:, v0:v1] if v1 > v0 else ts[:, :1], pe[:, v0:v1], ppe, cams)
        torch.save((shard.views, s, ps), os.path.join(out, "b%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _batch2_inputs():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import fake_ops
    from conftest import build_tiny_oracle, cam4, golden
    from test_engine_logic_cpu import MODS, hip_model
    for name in MODS + ["panfusion_amd.sharding"]:
        importlib.import_module(name).ops = fake_ops
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    model = hip_model(build_tiny_oracle())
    gen = torch.Generator().manual_seed(5)
    lat = torch.randn(2, 4, 4, 16, 16, generator=gen)                 # two different samples
    pano = torch.randn(2, 1, 4, 16, 32, generator=gen)
    cams = {k: torch.stack([v, v]) for k, v in cam4().items()}
    return model, (lat, pano, torch.full((2, 4), 981), t("prompt_embd"), t("pano_prompt_embd"), cams)


def test_sharded_forward_takes_a_multi_prompt_batch():
    """the sharded EPA used to raise for b != 1.  World 4, split 1 / 3 (group 0 owns the panorama): every rank's
    slice of a two-sample batch equals the single-process forward."""
    model, (lat, pano, ts, pe, ppe, cams) = _batch2_inputs()
    want_s, want_ps = model(lat, pano, ts, pe, ppe, cams)
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_batch2_worker, args=(4, _free_port(), out), nprocs=4, join=True)
        res = [torch.load(os.path.join(out, "b%d.pt" % r)) for r in range(4)]
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for (v0, v1), s, ps in res:
        assert s.shape[1] == v1 - v0 and rel(s, want_s[:, v0:v1]) < 1e-4
        if ps is not None:
            assert rel(ps, want_ps) < 1e-4
    assert sum(ps is not None for _, _, ps in res) == 2             # one panorama owner per CFG half
