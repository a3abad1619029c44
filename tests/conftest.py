import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def golden():
    """Vectors produced by running the reference (tests/golden/make_golden.py)."""
    arrays = np.load(os.path.join(GOLDEN, 'reference_vectors.npz'), allow_pickle=False)
    with open(os.path.join(GOLDEN, 'reference_vectors.json')) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return torch.device('cuda', 0)


def golden_graph_parts(arrays, name):
    """(features dict, nodes dict, edges) of a fixture graph."""
    feats = {'user': arrays[name + '_feat_user'], 'item': arrays[name + '_feat_item']}
    nu, ni = feats['user'].shape[0], feats['item'].shape[0]
    nodes = {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}
    return feats, nodes, arrays[name + '_edges']
