"""Host-side logic of the product that runs without a GPU: schedule, camera helpers,
weight packing (on the CPU device), API surface."""
import numpy as np
import torch

from conftest import TINY
from oracle import ddim as oddim
from oracle import geometry as G
from oracle import sd2_unet as U
from panfusion_amd import engine, pipeline
from panfusion_amd.external.Perspective_and_Equirectangular.e2p import _camera_lists
from panfusion_amd.models.pano import MultiViewBaseModel, WarpAttn
from panfusion_amd.models.pano.modules import camera_groups
from panfusion_amd.utils import pano as upano


def test_ddim_schedule_matches_oracle():
    a, b = pipeline.DDIMSchedule(), oddim.DDIM()
    for n in (50, 10, 3):
        assert a.set_timesteps(n) == b.set_timesteps(n).tolist()
        for t in a.timesteps:
            assert np.allclose(a.coefficients(t), [float(c) for c in b.coefficients(t)], rtol=0, atol=0)
    assert a.set_timesteps(50)[:2] == [981, 961] and a.timesteps[-1] == 1


def test_camera_samplers_match_oracle():
    a, b = upano.icosahedron_sample_camera(), G.icosahedron_cameras()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a, b = upano.horizon_sample_camera(8), G.horizon_cameras(8)
    assert np.array_equal(a[0], b[0])


def test_camera_list_broadcast():
    assert _camera_lists(5, 90, 10.0, 20.0) == ([90], [10.0], [20.0])
    f, u, v = _camera_lists(3, torch.tensor([90, 90, 90]), torch.tensor([1., 2., 3.]), [0., 0., 0.])
    assert f == [90, 90, 90] and u == [1.0, 2.0, 3.0]
    m, groups = camera_groups({"FoV": torch.full((4,), 90), "theta": torch.tensor([0., 1, 0, 1]),
                               "phi": torch.zeros(4)}, 2)
    assert m == 2 and groups[0] == groups[1]


def test_rotate_cameras_wraps():
    c = pipeline.rotate_cameras({"theta": torch.tensor([[300.0, -144.0]])}, 90.0)
    assert torch.equal(c["theta"], torch.tensor([[30.0, 306.0]]))


def test_module_tree_and_state_dict_layout():
    cfg = U.tiny_config(**TINY)
    unet, pano_unet = U.UNet2DConditionModel(**cfg), U.UNet2DConditionModel(**cfg)
    m = MultiViewBaseModel(unet, pano_unet)
    assert len(m.cp_blocks_encoder) == 3 and len(m.cp_blocks_decoder) == 3
    dims = [b.transformer.norm1.weight.shape[0] for b in [*m.cp_blocks_encoder, m.cp_blocks_mid, *m.cp_blocks_decoder]]
    assert dims == [64, 128, 256, 256, 256, 256, 128]         # MVGenModel.py:19-32 at width 64
    keys = set(m.cp_blocks_mid.state_dict())
    want = {"transformer.attn1.to_q.weight", "transformer.attn1.to_k.weight", "transformer.attn1.to_v.weight",
            "transformer.attn1.to_out.weight", "transformer.attn1.to_out.bias", "transformer.ff.net.0.proj.weight",
            "transformer.ff.net.0.proj.bias", "transformer.ff.net.2.weight", "transformer.ff.net.2.bias",
            "transformer.norm1.weight", "transformer.norm1.bias", "transformer.norm2.weight",
            "transformer.norm2.bias", "pe.freq_bands"}
    assert keys == want
    assert len(m.trainable_parameters) == 1 and m.trainable_parameters[0][1] == 1.0
    # zero-initialised output projections (identity at init, transformer.py:29-30,54-55)
    w = WarpAttn(64)
    assert float(w.transformer.attn1.to_out.weight.abs().sum()) == 0.0
    assert float(w.transformer.ff.net[2].weight.abs().sum()) == 0.0
    # PanoOnly construction shape
    po = MultiViewBaseModel(None, pano_unet)
    assert not hasattr(po, "cp_blocks_mid")


def test_weight_packing_layouts_and_lora_fold():
    cfg = U.tiny_config(**TINY)
    unet = U.UNet2DConditionModel(**cfg)
    unet.add_lora(4)
    U.init_synthetic(unet, 3)
    u = engine.pack_unet(unet, torch.device("cpu"), torch.float32)
    r = unet.down_blocks[1].resnets[0]
    p = u.down[1].resnets[0]
    assert p.w1.shape == (128, 9 * 64) and p.ws.shape == (128, 64)
    # [Cout, ky, kx, Cin] ordering
    assert torch.equal(p.w1.view(128, 3, 3, 64)[5, 1, 2], r.conv1.weight[5, :, 1, 2])
    a = unet.down_blocks[0].attentions[0].transformer_blocks[0].attn1
    pa = u.down[0].attns[0].attn1
    want_q = a.to_q.weight + a.to_q.lora_layer.up.weight @ a.to_q.lora_layer.down.weight
    assert torch.allclose(pa.wqk[:64], want_q) and pa.wqk.shape == (128, 64)
    # concatenated time-embedding projection: offsets tile the table
    offs = [x.temb_off for b in u.down for x in b.resnets] + [x.temb_off for x in u.mid.resnets] + \
           [x.temb_off for b in u.up for x in b.resnets]
    assert offs[0] == 0 and sorted(offs) == offs and u.w_temb.shape == (u.temb_total, 256)
    assert u.w_conv_in.shape == (3, 3, 4, 64) and u.w_conv_out.shape == (4, 3, 3, 64)


def test_product_never_imports_oracle():
    import os
    import re
    from conftest import ROOT
    for dp, _, files in os.walk(os.path.join(ROOT, "panfusion_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
