"""Dropout masks of the device's training kernel, restated on the CPU -- test infrastructure.

The reference applies ``Dropout(rate)`` to the head input and to the neighbor input of every aggregator call
(``models.py:152,165``), drawn by TensorFlow's generator.  The device draws them from Philox, addressed like its
sampling draws so that they can be restated here: the four columns ``4*c4 .. 4*c4+3`` of row ``row`` of input
``site`` of walk ``w`` are kept where the four words of

    philox4x32_10(counter = (w lo, w hi, 0x40000000 | site, row * 64 + c4), key = dropout seed)

are ``>= floor(rate * 2**32)``, and kept values are multiplied by ``1 / (1 - rate)`` (both in float32).  Sites of a
two-hop walk: 0 the hop-2 rows (row = j * n1 + k), 1 the hop-1 rows as heads, 2 the hop-1 rows as the root's
neighbors, 3 the root's row, 4 the layer-1 inputs (row 0 the root's layer-0 output, row 1 + j hop-1 head j's).
"""
import numpy as np
import torch

from . import philox

SITE_TAG = 0x40000000


def keep_mask(walk_ids, site, rows, cols, rate, seed):
    """bool [len(walk_ids), len(rows), cols]."""
    walk = np.asarray(walk_ids, dtype=np.uint64)[:, None, None]
    rows = np.asarray(rows, dtype=np.uint64)[None, :, None]
    c4 = np.arange(cols // 4, dtype=np.uint64)[None, None, :]
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    words = philox.philox4x32_10(walk & np.uint64(0xFFFFFFFF), walk >> np.uint64(32), SITE_TAG | site,
                                 rows * np.uint64(64) + c4, seed & 0xFFFFFFFF, seed >> 32)
    thr = max(1, int(float(np.float32(rate)) * 4294967296.0))
    keep = np.stack([w >= np.uint32(thr) for w in words], axis=-1)       # [..., c4, 4]
    return keep.reshape(keep.shape[0], keep.shape[1], cols)


def two_hop(walk_ids, batch, n0, n1, rate, seed):
    """The hook ``oracle.model.forward(dropout=...)`` calls as ``hook(layer, slot, role, x)``: x is the head input
    (B, heads, F) or the neighbor input (B, heads, n, F) of aggregator call ``slot`` of ``layer``."""
    walk_ids = np.asarray(walk_ids)
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))

    def hook(layer, slot, role, x):
        walks = walk_ids[:batch] if slot % 2 == 0 else walk_ids[batch:]
        cols = x.shape[-1]
        if layer == 0 and slot < 2:
            site, rows = (3, [0]) if role == 'head' else (2, range(n0))
        elif layer == 0:
            site, rows = (1, range(n0)) if role == 'head' else (0, range(n0 * n1))
        else:
            site, rows = (4, [0]) if role == 'head' else (4, range(1, n0 + 1))
        keep = torch.from_numpy(keep_mask(walks, site, list(rows), cols, rate, seed)).reshape(x.shape)
        return torch.where(keep, x * scale, torch.zeros_like(x))

    return hook
