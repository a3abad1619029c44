"""The C ABI used from a plain C++ program (tests/abi/abi_smoke.cpp, built by
__graft_entry__.build()): adjacency build, tape-replayed walks, gather + mean +
projection, each checked inside the program against host arithmetic -- needs a GPU."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_library_works_without_python_or_torch(cuda):
    exe = os.path.join(ROOT, 'tests', 'abi', 'abi_smoke')
    if not os.path.exists(exe):
        pytest.skip("tests/abi/abi_smoke is built by __graft_entry__.build()")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'abi smoke ok' in out.stdout
    assert 'gs_csr_build: 0 mismatches' in out.stdout and 'gs_sample (tape replay): 0 mismatches' in out.stdout
