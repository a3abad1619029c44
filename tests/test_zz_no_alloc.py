"""The entry point that allocates nothing (``gs_link_head_mse_fwd_bwd_static``) against the
entry point it is a variant of (``gs_link_head_mse_fwd_bwd``, caller's workspace), straight
through the C ABI -- needs a GPU.  Two calls in a row, the second smaller, so that whatever
the first call left in the library's static scratch must not leak into the second; results
are compared bit for bit (same kernels, same order of additions)."""
import ctypes

import pytest
import torch

from graphsage_amd import _lib

pytestmark = pytest.mark.gpu


def _head(lib, static, eu, ei, kernel, bias, y):
    b, e = eu.shape
    dev = eu.device
    out = {'pred': torch.full((b,), -7.0, device=dev), 'loss': torch.full((1,), -7.0, device=dev),
           'g_eu': torch.full((b, e), -7.0, device=dev), 'g_ei': torch.full((b, e), -7.0, device=dev),
           'g_kernel': torch.full((2 * e,), -7.0, device=dev), 'g_bias': torch.full((1,), -7.0, device=dev)}
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [p(eu), p(ei), b, e, p(kernel), p(bias), p(y), p(out['pred']), p(out['loss']), p(out['g_eu']), p(out['g_ei']),
            p(out['g_kernel']), p(out['g_bias'])]
    if static:
        rc = lib.gs_link_head_mse_fwd_bwd_static(*args, stream)
    else:
        need = lib.gs_link_head_workspace(b, e)
        ws = torch.full((need // 4,), float('nan'), device=dev)
        rc = lib.gs_link_head_mse_fwd_bwd(*args, p(ws), need, stream)
    assert rc == 0, lib.gs_last_error()
    torch.cuda.synchronize()
    return out


def test_static_scratch_head_equals_workspace_head_on_two_calls_in_a_row(cuda):
    lib = _lib.load()
    gen = torch.Generator(device=cuda).manual_seed(5)
    for b, e in [(4096, 256), (100, 16), (4096, 32), (1, 1)]:          # large first, then smaller ones
        eu = torch.randn((b, e), device=cuda, generator=gen)
        ei = torch.randn((b, e), device=cuda, generator=gen)
        kernel = torch.randn((2 * e,), device=cuda, generator=gen) * 0.3
        bias = torch.randn((1,), device=cuda, generator=gen)
        y = (torch.rand((b,), device=cuda, generator=gen) > 0.5).float()
        want = _head(lib, False, eu, ei, kernel, bias, y)
        got = _head(lib, True, eu, ei, kernel, bias, y)
        for name in want:
            assert torch.equal(got[name], want[name]), (b, e, name)
        assert float(want['loss']) > 0 and bool(torch.isfinite(want['g_kernel']).all())
        # and against plain torch arithmetic in float64
        z = torch.cat([eu, ei], 1).double() @ kernel.double() + bias.double()
        pr = torch.sigmoid(z)
        assert float((got['pred'].double() - pr).abs().max()) < 1e-6
        assert abs(float(got['loss']) - float(((pr - y.double()) ** 2).mean())) < 1e-6


def test_static_scratch_head_refuses_what_its_array_cannot_hold(cuda):
    lib = _lib.load()
    z = torch.zeros(2 * 257 * 4, device=cuda)
    p = ctypes.c_void_p(z.data_ptr())
    rc = lib.gs_link_head_mse_fwd_bwd_static(p, p, 4, 257, p, p, p, p, p, p, p, p, p, None)
    assert rc == -3 and b'static scratch' in lib.gs_last_error()      # GS_ELIMIT
    assert lib.gs_link_head_mse_fwd_bwd_static(p, p, 0, 8, p, p, p, p, p, p, p, p, p, None) == -1
    assert lib.gs_link_head_mse_fwd_bwd_static(None, p, 4, 8, p, p, p, p, p, p, p, p, p, None) == -1
