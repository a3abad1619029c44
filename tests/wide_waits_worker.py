"""Runs the wide fused layer over a mix of tile counts, child counts (full chunks, a last chunk of one child, a
single child), zero rows and saved copies, and writes every output to ``--out``.  Started twice by
tests/test_gpu_parity.py::test_wide_kernel_wait_counts -- once on the product library, once with GSAGE_LIB naming
the build whose hand-counted waits are replaced by waits for every load -- never collected by pytest."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from graphsage_amd import kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(11)
    outs = {}
    case = 0
    for fs, fn, h in [(256, 256, 256), (64, 96, 256)]:
        n_self, n_neigh, off_s, off_n = 4000, 3000, 5, 5000
        t_self = torch.from_numpy(rng.standard_normal((n_self, fs)).astype(np.float32)).to(dev)
        t_neigh = torch.from_numpy(rng.standard_normal((n_neigh, fn)).astype(np.float32)).to(dev)
        ws = torch.from_numpy(rng.standard_normal((fs, h // 2)).astype(np.float32) * 0.05).to(dev)
        wn = torch.from_numpy(rng.standard_normal((fn, h // 2)).astype(np.float32) * 0.05).to(dev)
        bias = torch.from_numpy(rng.standard_normal(h).astype(np.float32) * 0.1).to(dev)
        for mix in ([(700 * 128 + 3, 5), (40, 1), (129, 6)], [(300 * 128, 10), (5, 25), (1000, 11)],
                    [(513, 32), (128 * 260, 1), (77, 3)], [(1, 1)], [(128 * 300 + 1, 6)]):
            segs, rows = [], 0
            for k, (n_heads, nc) in enumerate(mix):
                heads = rng.integers(0, n_self, n_heads).astype(np.int32) + off_s
                kids = rng.integers(0, n_neigh, n_heads * nc).astype(np.int32) + off_n
                if (case + k) % 2:                      # zero rows in some tiles only, heads outside the table too
                    kids[:len(kids) // 2:53] = -1
                    heads[::91] = off_s + n_self + 3
                segs.append({'head_ids': torch.from_numpy(heads).to(dev), 'child_ids': torch.from_numpy(kids).to(dev),
                             'n_heads': n_heads, 'n_child': nc})
                rows += n_heads
            sv = {}
            if case % 2 == 0:
                sv = {'save_self': torch.zeros((rows, fs), device=dev), 'save_mean': torch.zeros((rows, fn), device=dev)}
            for pieces in (True, False):                # both forms of the product share the producers
                got = kernels.sage_fused(t_self, off_s, t_neigh, off_n, segs, ws, wn, bias, 'relu', exact_pieces=pieces, **sv)
                torch.cuda.synchronize()
                outs['out%d_%d' % (case, pieces)] = got.cpu()
                for k, v in sv.items():
                    outs['%s%d_%d' % (k, case, pieces)] = v.cpu()
            case += 1
    torch.save(outs, args.out)
    print("wide_waits_worker: %d cases on %s" % (case, kernels._lib.LIB_PATH))


if __name__ == '__main__':
    main()
