"""CLIP text encoding on the HIP kernels (reference PanoGenerator.py:197-211,
PanFusion.py:45-62,134-138) against transformers' own CLIPTextModel on the CPU (fp32).  Needs an MI355X: `-m gpu`."""
import time

import pytest
import torch

from conftest import rel_l2
from test_engine_logic_cpu import tiny_clip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_embed_tokens_kernel():
    from panfusion_amd import ops
    g = torch.Generator().manual_seed(0)
    tok, pos = torch.randn(500, 64, generator=g).to(DEV), torch.randn(77, 64, generator=g).to(DEV)
    ids = torch.randint(0, 500, (3, 77), generator=g).to(DEV)
    for dt in (torch.float32, torch.float16):
        out = ops.embed_tokens(ids, tok, pos, 80, dt)
        want = torch.zeros(3, 80, 64, device=DEV)
        want[:, :77] = tok[ids] + pos
        assert out.dtype == dt and torch.equal(out, want.to(dt))


@pytest.mark.parametrize("dtype,precision,tol", [(torch.float16, "mixed", 1e-3), (torch.float16, "fast", 4e-3),
                                                 (torch.bfloat16, "fast", 3e-2)])
def test_text_encoder_tiny_vs_transformers(dtype, precision, tol):
    from panfusion_amd.text_encoder import TextEncoder
    m = tiny_clip(layers=4)
    ids = torch.randint(1, 990, (5, 77), generator=torch.Generator().manual_seed(2))
    ids[:, 0], ids[0, 9:], ids[3, 60:] = 998, 999, 999
    with torch.no_grad():
        want = m(ids)[0]
    got = TextEncoder(m, compute_dtype=dtype, precision=precision).encode_ids(ids.to(DEV))
    err = rel_l2(got.cpu(), want)
    print("text encoder tiny %s/%s: rel-L2 %.3e" % (dtype, precision, err))
    assert got.shape == (5, 77, 128) and err <= tol


def test_text_encoder_sd2_shape_vs_transformers():
    """The SD-2 text encoder's real shape (OpenCLIP-H as shipped with stable-diffusion-2-base: width 1024, 16 heads
    of 64, 23 layers, MLP 4096, vocabulary 49408), random weights, the 22 prompts of one sample (20 views +
    panorama + null): HIP fp16 mixed vs transformers fp32 on the CPU."""
    from transformers import CLIPTextConfig, CLIPTextModel
    from panfusion_amd.text_encoder import TextEncoder
    torch.manual_seed(0)
    cfg = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=23,
                         vocab_size=49408, max_position_embeddings=77, hidden_act="gelu", projection_dim=512)
    m = CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    ids = torch.randint(1, 49000, (22, 77), generator=torch.Generator().manual_seed(3))
    ids[:, 0] = 49406
    for b in range(22):
        ids[b, 5 + 3 * b:] = 49407
    with torch.no_grad():
        want = m(ids)[0]
    enc = TextEncoder(m, compute_dtype=torch.float16)
    got = enc.encode_ids(ids.to(DEV))
    err = rel_l2(got.cpu(), want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        enc.encode_ids(ids.to(DEV))
    torch.cuda.synchronize()
    print("text encoder SD-2 shape, 22 prompts, fp16 %s: rel-L2 %.3e, %.2f ms per call (eager launches)"
          % (enc.precision, err, (time.perf_counter() - t0) / 5 * 1e3))
    assert got.shape == (22, 77, 1024) and err <= 1e-3
