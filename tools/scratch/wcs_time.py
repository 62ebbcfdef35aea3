"""Time of pf_weighted_colsum on the training step's shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from panfusion_amd import ops
for T, C, R in [(20480, 320, 4), (20480, 960, 12), (5120, 640, 4), (5120, 1920, 12), (1280, 1280, 4), (1280, 3840, 12), (8192, 320, 4), (8192, 960, 12), (2560, 1024, 8), (128, 1024, 8)]:
    x = torch.randn(T, C, device="cuda").half()
    w = torch.randn(R, T, device="cuda")
    for _ in range(3):
        ops.weighted_colsum(x, w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.weighted_colsum(x, w)
    torch.cuda.synchronize()
    print("T %6d C %5d R %2d   %7.1f us" % (T, C, R, (time.perf_counter() - t0) / 50 * 1e6))
