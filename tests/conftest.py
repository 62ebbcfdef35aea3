import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota (the GPU box exposes all
    host cores to os.cpu_count() but grants a fraction; torch's default thread count then
    oversubscribes the oracle by an order of magnitude)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    workers = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "1") or 1)
    torch.set_num_threads(max(1, min(16, _usable_cores() // max(1, workers))))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


TINY = dict(width=64, cross_attention_dim=128, heads=(1, 2, 4, 4), groups=32)


def build_tiny_oracle(seed=11, lora_rank=4):
    """Same construction as tools/make_golden.py:build_tiny_models, on the oracle classes."""
    from oracle import mvgen as MV
    from oracle import sd2_unet as U
    cfg = U.tiny_config(**TINY)
    unet, pano_unet = U.UNet2DConditionModel(**cfg), U.UNet2DConditionModel(**cfg)
    unet.add_lora(lora_rank)                    # (a hyperparameter of the reference: PanoGenerator.py:73)
    pano_unet.add_lora(lora_rank)
    U.init_synthetic(unet, seed)
    U.init_synthetic(pano_unet, seed + 1)
    model = MV.DualBranchDenoiser(unet, pano_unet, None, None, True)
    U.init_synthetic(model.cp_blocks_encoder, seed + 2)
    U.init_synthetic(model.cp_blocks_mid, seed + 3)
    U.init_synthetic(model.cp_blocks_decoder, seed + 4)
    MV.randomize_epa(model, seed + 5)
    return model


def cam4():
    return {"FoV": torch.full((4,), 90), "theta": torch.tensor([0., 90, 180, 270], dtype=torch.float64),
            "phi": torch.tensor([0., 10, -20, 45], dtype=torch.float64)}


def unpair(p):
    """Split-precision pair [.., 2C] of the kernels (per block of 32 channels [hi(32) | lo(32)]) -> (hi, lo) [.., C]."""
    C2 = p.shape[-1]
    q = p.reshape(*p.shape[:-1], C2 // 64, 2, 32)
    return q[..., 0, :].reshape(*p.shape[:-1], C2 // 2), q[..., 1, :].reshape(*p.shape[:-1], C2 // 2)
