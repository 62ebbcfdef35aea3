"""How long does re-packing a training ControlNet take (MultiViewBaseModel.packed after its parameters moved)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from panfusion_amd.models.sd2_unet_params import SD2_BASE
dev = torch.device("cuda")
model = bench.build_model(dev, torch.float16, dict(SD2_BASE), layout_cond=True)
model.packed("pano_cn", dev)
params = list(model.pano_cn.parameters())
for rep in range(3):
    with torch.no_grad():
        for p in params:
            p.add_(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.packed("pano_cn", dev)
    torch.cuda.synchronize()
    print("ControlNet re-pack %.2f ms" % ((time.perf_counter() - t0) * 1e3))
