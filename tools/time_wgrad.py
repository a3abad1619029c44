#!/usr/bin/env python3
"""Times the weight-gradient half of gs_sage_project_bwd on a wide layer (256-wide inputs, 256 outputs).
python tools/time_wgrad.py [rows] [act: relu/linear];  GSAGE_LIB selects the library build."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import kernels

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_130_496
act = sys.argv[2] if len(sys.argv) > 2 else 'relu'
f, h = 256, 256
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
xs = torch.randn((m, f), device=dev, generator=g)
xm = torch.randn((m, f), device=dev, generator=g)
out = torch.randn((m, h), device=dev, generator=g).clamp_(min=0)
go = torch.randn((m, h), device=dev, generator=g)
ws = torch.randn((f, h // 2), device=dev, generator=g)
wn = torch.randn((f, h // 2), device=dev, generator=g)
gws, gwn, gb = torch.empty_like(ws), torch.empty_like(wn), torch.empty(h, device=dev)
wsp = torch.empty(kernels.project_bwd_workspace_floats(f, f, h, m), device=dev)


def run():
    kernels.sage_project_bwd(xs, xm, ws, wn, act, out if act == 'relu' else None, go, need_x_self=False, need_x_mean=False,
                             g_w_self=gws, g_w_neigh=gwn, g_bias=gb, workspace=wsp)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
flops = 2.0 * m * (2 * f) * (h // 2)
byts = m * (2 * f + h * (2 if act == 'relu' else 1)) * 4
ref = (xs.double().t() @ (go[:, :h // 2] * (out[:, :h // 2] > 0 if act == 'relu' else 1)).double())
err = float((gws.double() - ref).abs().max() / ref.abs().max())
print("lib %s  rows %d %s: %.1f us  %.1f TFLOP/s  %.2f TB/s (operands once)  rel err %.1e" % (
    os.environ.get('GSAGE_LIB', 'default'), m, act, us, flops / us * 1e-6, byts / us * 1e-6, err))
