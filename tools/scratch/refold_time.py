"""How long does the per-step LoRA re-fold take (MultiViewBaseModel.refold_lora + the training packs' rebuild)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from panfusion_amd import engine, train_engine
from panfusion_amd.models.sd2_unet_params import SD2_BASE
dev = torch.device("cuda")
model = bench.build_model(dev, torch.float16, dict(SD2_BASE))
model.differentiable = True
for which in ("unet", "pano_unet"):
    model.packed(which, dev)
params = model.trainable_tensors()
def bump():
    with torch.no_grad():
        for p in params:
            p.add_(0)
for rep in range(3):
    bump()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.refold_lora()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for (which, *_), u in model._packed.items():
        for t in engine.all_transformers(u):
            train_engine.transformer_train(t, dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("refold_lora %.2f ms   training packs (transposes, cats) %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
