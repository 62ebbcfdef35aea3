"""Training-resolution forward with FRESH random cameras every call (the reference's training sampler, PanoDataset.py:99-101
random_sample_camera(20)) vs the same cameras every call: what the per-step EPA table builds cost."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from panfusion_amd.models.sd2_unet_params import SD2_BASE
from panfusion_amd.utils.pano import icosahedron_sample_camera, random_sample_camera
dev = torch.device("cuda")
model = bench.build_model(dev, torch.float16, dict(SD2_BASE))
m = 20
g = lambda s: torch.Generator().manual_seed(s)
lat, pano = torch.randn(1, m, 4, 32, 32, generator=g(0)).to(dev), torch.randn(1, 1, 4, 64, 128, generator=g(1)).to(dev)
pr, ppr = torch.randn(1, m, 77, 1024, generator=g(4)).to(dev), torch.randn(1, 1, 77, 1024, generator=g(5)).to(dev)
t = torch.full((1, m), 500, device=dev)
def cams(th, ph):
    return {"FoV": torch.full((1, m), 90), "theta": torch.tensor(np.degrees(th), dtype=torch.float64)[None], "phi": torch.tensor(np.degrees(ph), dtype=torch.float64)[None]}
fixed = cams(*icosahedron_sample_camera())
with torch.no_grad():
    for name, fresh in (("fixed cameras", False), ("fresh random cameras", True), ("fixed cameras", False)):
        for _ in range(2):
            model(lat, pano, t, pr, ppr, fixed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(6):
            model(lat, pano, t, pr, ppr, cams(*random_sample_camera(m)) if fresh else fixed)
        torch.cuda.synchronize()
        print("%-24s %.2f ms per forward" % (name, (time.perf_counter() - t0) / 6 * 1e3))
