"""Times one wide fused layer (gs_sage_fused_fwd on 256-wide inputs) on synthetic ids.
python tools/time_wide.py [heads] [n_child] [table_rows] [save: 0/1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from graphsage_amd import kernels

heads = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
n_child = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
save = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
f, h = 256, 256
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
t_self = torch.randn((100_000, f), device=dev, generator=g)
t_neigh = torch.randn((rows, f), device=dev, generator=g)
hid = torch.randint(0, t_self.shape[0], (heads,), device=dev, generator=g, dtype=torch.int32)
cid = torch.randint(0, rows, (heads * n_child,), device=dev, generator=g, dtype=torch.int32)
ws = torch.randn((f, h // 2), device=dev, generator=g) * 0.05
wn = torch.randn((f, h // 2), device=dev, generator=g) * 0.05
bias = torch.zeros(h, device=dev)
out = torch.empty((heads, h), device=dev)
sv = {'save_self': torch.empty((heads, f), device=dev), 'save_mean': torch.empty((heads, f), device=dev)} if save else {}
call = kernels.FusedLayerCall(t_self, 0, t_neigh, 0, [{'head_ids': hid, 'child_ids': cid, 'n_heads': heads, 'n_child': n_child}],
                              ws, wn, bias, 'relu', False, out, **sv)
for _ in range(3):
    call.run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t = kernels.LaunchTimer(); t.arm(); call.run(); ts.append(t)
torch.cuda.synchronize()
us = sorted(t.elapsed_us() for t in ts)[len(ts) // 2]
flops = heads * f * (h // 2) * 2 * 2
byts = heads * (1 + n_child) * f * 4 + heads * h * 4
print("%sheads %d n_child %d table %d rows: %.1f us  %.1f TFLOP/s  %.2f TB/s algorithmic" % ("saving, " if save else "", heads, n_child, rows, us, flops / us * 1e-6, byts / us * 1e-6))
