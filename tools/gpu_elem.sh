#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "groupnorm or layernorm or pointwise" 2>&1 | tail -n 4 | cut -c1-300
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from panfusion_amd import ops
dev='cuda'
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(0.01*2.4e9)); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
for rows, C in ((163840, 320), (40960, 640), (10240, 1280)):
    x=torch.randn(rows, C, device=dev).to(torch.bfloat16); g=torch.ones(C, device=dev); b=torch.zeros(C, device=dev)
    us=timeit(lambda: ops.layernorm(x, g, b)); print("layernorm %7d x %4d  %7.1f us  %.2f TB/s" % (rows, C, us, rows*C*4/us/1e6))
    n=40; hw=rows//n
    sc=torch.ones(n, C, device=dev); sh=torch.zeros(n, C, device=dev)
    us=timeit(lambda: ops.scale_shift_act(x.view(n,hw,C), None, n, hw, sc, sh, 1)); print("gn-apply  %7d x %4d  %7.1f us  %.2f TB/s" % (rows, C, us, rows*C*4/us/1e6))
    gg=torch.ones(C, device=dev); bb=torch.zeros(C, device=dev)
    us=timeit(lambda: ops.groupnorm_scale_shift(x.view(n,hw,C), None, n, hw, 32, 1e-5, gg, bb)); print("gn-stats  %7d x %4d  %7.1f us  %.2f TB/s (read only)" % (rows, C, us, rows*C*2/us/1e6))
PY
