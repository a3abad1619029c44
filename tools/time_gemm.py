#!/usr/bin/env python3
"""Times gs_sage_project_fwd on a wide layer (default: the largest projection of the 3-hop
256-wide workload); GSAGE_LIB selects the library build."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import kernels

m, fs, fn, h = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (282624, 256, 256, 256)))
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
xs = torch.randn(m, fs, device=dev, generator=g)
xm = torch.randn(m, fn, device=dev, generator=g)
ws = torch.randn(fs, h // 2, device=dev, generator=g) * 0.05
wn = torch.randn(fn, h // 2, device=dev, generator=g) * 0.05
b = torch.randn(h, device=dev, generator=g)
out = torch.empty(m, h, device=dev)
for _ in range(5):
    kernels.sage_project(xs, xm, ws, wn, b, 'relu', out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for _ in range(n):
    kernels.sage_project(xs, xm, ws, wn, b, 'relu', out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
want = torch.relu(torch.cat([xs[:512] @ ws, xm[:512] @ wn], dim=1) + b)
err = float((out[:512] - want).abs().max())
print("lib %s  M=%d K=%d+%d H=%d: %.1f us  %.1f TFLOP/s  max|err| %.2e" % (
    os.path.basename(os.environ.get('GSAGE_LIB', 'default')), m, fs, fn, h, us,
    2.0 * m * (fs + fn) * (h // 2) / us / 1e6, err))
