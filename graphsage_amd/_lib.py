"""ctypes binding of ``libgsage.so`` (the HIP kernels + C ABI in ``csrc/``).

There is no fallback: if the library is missing or a call fails, an exception
is raised.  ``torch`` is imported before the library is opened so that both use
the one HIP runtime torch has already loaded (same SONAME), which is what makes
torch's device pointers and stream handles valid inside the library.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GSAGE_LIB') or os.path.join(_HERE, 'libgsage.so')   # GSAGE_LIB: another build of the same ABI

GS_OK = 0
GS_MAX_HOPS = 4
ACT = {'linear': 0, 'relu': 1}

_c = ctypes
_vp, _i32, _i64, _u64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_uint64, _c.c_size_t

# name -> (restype, argtypes); mirrors include/gsage.h declaration by declaration
SIGNATURES = {
    'gs_version': (_i32, []),
    'gs_last_error': (_c.c_char_p, []),
    'gs_timer_create': (_i32, [_c.POINTER(_vp)]),
    'gs_timer_arm': (_i32, [_vp]),
    'gs_timer_elapsed_us': (_i32, [_vp, _c.POINTER(_c.c_float)]),
    'gs_timer_destroy': (_i32, [_vp]),
    'gs_sample': (_i32, [_vp, _vp, _i64, _vp, _i64, _c.POINTER(_c.c_int32), _i32, _u64, _u64, _vp, _vp,
                         _c.POINTER(_vp), _vp, _vp]),
    'gs_sample_cyclic': (_i32, [_vp, _vp, _i64, _vp, _i64, _c.POINTER(_c.c_int32), _i32, _vp, _u64, _vp, _vp, _i64,
                                _c.POINTER(_vp), _vp, _vp]),
    'gs_csr_build_workspace': (_i32, [_i64, _i64, _i64, _c.POINTER(_sz)]),
    'gs_csr_build': (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    'gs_gather_rows': (_i32, [_vp, _i64, _i32, _i64, _vp, _i64, _vp, _vp]),
    'gs_gather_mean_fwd': (_i32, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _vp, _vp]),
    'gs_gather_max_fwd': (_i32, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _vp, _vp]),
    'gs_group_max_bwd': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    'gs_gather_mean_bwd': (_i32, [_vp, _vp, _i64, _i32, _i32, _i64, _i64, _vp, _vp]),
    'gs_sage_project_fwd': (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    'gs_sage_project_gather_fwd': (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32,
                                          _vp, _vp]),
    'gs_sage_project_bwd_workspace': (_sz, [_i32, _i32, _i32, _i64]),
    'gs_sage_project_bwd': (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gs_l2_normalize_fwd': (_i32, [_vp, _i64, _i32, _vp, _vp]),
    'gs_l2_normalize_bwd': (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
}



class FusedSegment(_c.Structure):
    """gs_fused_segment of include/gsage.h."""
    _fields_ = [('head_ids', _vp), ('child_ids', _vp), ('n_heads', _i64), ('self_row0', _i64),
                ('child_row0', _i64), ('n_child', _c.c_int32)]


SIGNATURES['gs_sage_fused_supported'] = (_i32, [_i32, _i32, _i32, _i32])
SIGNATURES['gs_sage_fused_fwd'] = (_i32, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i64,
                                          _c.POINTER(FusedSegment), _i32, _vp, _vp, _vp, _i32, _i32, _i32,
                                          _vp, _vp, _vp, _vp])
SIGNATURES['gs_sage_fused_workspace_bytes'] = (_sz, [_i32, _i32, _i32, _i32])
SIGNATURES['gs_sage_fused_fwd_ws'] = (_i32, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i64,
                                             _c.POINTER(FusedSegment), _i32, _vp, _vp, _vp, _i32, _i32, _i32,
                                             _vp, _vp, _vp, _vp, _sz, _vp])



class FusedSampling(_c.Structure):
    """gs_fused_sampling of include/gsage.h."""
    _fields_ = [('indptr', _vp), ('indices', _vp), ('n_nodes', _i64), ('rng_seed', _u64), ('walk_id_base', _u64),
                ('walk_ids', _vp), ('replay', _vp), ('tape_len', _i64), ('err_flag', _vp)]


class FusedDraw(_c.Structure):
    """gs_fused_draw of include/gsage.h."""
    _fields_ = [('hop', _c.c_int32), ('heads_per_walk', _c.c_int32), ('tape_off', _c.c_int32), ('walk0', _i64),
                ('out_child_ids', _vp)]


SIGNATURES['gs_sage_fused_sampled_fwd'] = (_i32, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i64,
                                                  _c.POINTER(FusedSegment), _i32, _vp, _vp, _vp, _i32, _i32, _i32,
                                                  _vp, _vp, _vp, _c.POINTER(FusedSampling), _c.POINTER(FusedDraw), _vp])


class WgradProblem(_c.Structure):
    """gs_wgrad_problem of include/gsage.h."""
    _fields_ = [('x_self', _vp), ('x_mean', _vp), ('out', _vp), ('g_out', _vp), ('g_W_self', _vp), ('g_W_neigh', _vp),
                ('g_bias', _vp), ('workspace', _vp), ('workspace_bytes', _sz),
                ('self_ids', _vp), ('self_off', _i64), ('self_rows', _i64)]


SIGNATURES['gs_sage_project_wgrad_pair'] = (_i32, [_c.POINTER(WgradProblem), _c.POINTER(WgradProblem),
                                                   _i32, _i32, _i64, _i32, _i32, _vp])

class SageWeights(_c.Structure):
    """gs_sage_weights of include/gsage.h."""
    _fields_ = [('w_self', _vp), ('w_neigh', _vp), ('bias', _vp)]


class Sage2Saved(_c.Structure):
    """gs_sage2_saved of include/gsage.h."""
    _fields_ = [('x_self', _vp * 2), ('x_mean', _vp * 2), ('h1', _vp * 2), ('x_mean1', _vp * 2), ('z', _vp * 2),
                ('self_ids', _vp * 2), ('h1_mask', _vp * 2), ('dropout_rate', _c.c_float), ('dropout_seed', _u64)]


class Sage2Subtrees(_c.Structure):
    """gs_sage2_subtrees of include/gsage.h."""
    _fields_ = [('roots_per_walk', _i32), ('tape_len', _i64), ('tape_off_hop2', _i64), ('tape_off_hop3', _i64),
                ('out_h0', _vp), ('seeds', _vp), ('out_hop1', _vp),
                ('sv_x_mean0_hop2', _vp * 2), ('sv_mask_hop2', _vp * 2), ('sv_x_mean0_hop1', _vp * 2),
                ('sv_x_mean1_hop1', _vp * 2)]


class Sage3SeedsSaved(_c.Structure):
    """gs_sage3_seeds_saved of include/gsage.h."""
    _fields_ = [('x_mean0', _vp * 2), ('h0', _vp * 2), ('x_mean1', _vp * 2), ('h1', _vp * 2), ('x_mean2', _vp * 2),
                ('z', _vp * 2)]


class Sage3BwdHop1(_c.Structure):
    """gs_sage3_bwd_hop1 of include/gsage.h."""
    _fields_ = [('h0', _vp * 2), ('h1', _vp * 2), ('x_mean0', _vp * 2), ('x_mean1', _vp * 2), ('ids', _vp * 2)]


class Sage3BwdHop2(_c.Structure):
    """gs_sage3_bwd_hop2 of include/gsage.h."""
    _fields_ = [('h0', _vp * 2), ('mask', _vp * 2), ('x_mean0', _vp * 2), ('ids', _vp * 2)]


class Sage2L0Mask(_c.Structure):
    """gs_sage2_l0_mask of include/gsage.h."""
    _fields_ = [('h1', _vp * 2), ('h1_mask', _vp * 2), ('input_scale', _c.c_float)]


class Sage2BwdAll(_c.Structure):
    """gs_sage2_bwd_all of include/gsage.h."""
    _fields_ = [('z', _vp * 2), ('emb', _vp * 2), ('g_emb', _vp * 2),
                ('targets', _vp), ('head_kernel', _vp), ('head_bias', _vp),
                ('w1_self', _vp * 2), ('w1_neigh', _vp * 2),
                ('h1', _vp * 2), ('x_mean1', _vp * 2), ('h1_mask', _vp * 2),
                ('x_self', _vp * 2), ('x_mean', _vp * 2), ('self_ids', _vp * 2),
                ('self_off', _i64 * 2), ('self_rows', _i64 * 2),
                ('input_scale', _c.c_float), ('pred', _vp),
                ('g_w0_self', _vp * 2), ('g_w0_neigh', _vp * 2), ('g_b0', _vp * 2),
                ('g_w1_self', _vp * 2), ('g_w1_neigh', _vp * 2), ('g_b1', _vp * 2),
                ('g_head_kernel', _vp), ('g_head_bias', _vp), ('loss', _vp)]


class AdamUpdate(_c.Structure):
    """gs_adam_update of include/gsage.h."""
    _fields_ = [('grads', _vp), ('params', _vp), ('m', _vp), ('v', _vp), ('n', _i64), ('lr_t_device', _vp),
                ('lr_t', _c.c_float), ('beta_1', _c.c_float), ('beta_2', _c.c_float), ('epsilon', _c.c_float),
                ('grad_scale', _c.c_float)]


class Sage3Bwd(_c.Structure):
    """gs_sage3_bwd of include/gsage.h."""
    _fields_ = [('z', _vp * 2), ('emb', _vp * 2), ('g_emb', _vp * 2),
                ('targets', _vp), ('head_kernel', _vp), ('head_bias', _vp),
                ('w2_self', _vp * 2), ('w2_neigh', _vp * 2), ('w1_self', _vp * 2), ('w1_neigh', _vp * 2),
                ('h0', _vp * 2), ('h1', _vp * 2), ('x_mean0', _vp * 2), ('x_mean1', _vp * 2), ('x_mean2', _vp * 2),
                ('feat', _vp * 2), ('feat_off', _i64 * 2), ('feat_rows', _i64 * 2), ('row_ids', _vp * 2),
                ('pred', _vp),
                ('g_w0_self', _vp * 2), ('g_w0_neigh', _vp * 2), ('g_b0', _vp * 2),
                ('g_w1_self', _vp * 2), ('g_w1_neigh', _vp * 2), ('g_b1', _vp * 2),
                ('g_w2_self', _vp * 2), ('g_w2_neigh', _vp * 2), ('g_b2', _vp * 2),
                ('g_head_kernel', _vp), ('g_head_bias', _vp), ('loss', _vp),
                ('hop1', _c.POINTER(Sage3BwdHop1)), ('hop2', _c.POINTER(Sage3BwdHop2))]


SIGNATURES['gs_sage3_backward_all_supported'] = (_i32, [_i32] * 7)
SIGNATURES['gs_sage3_backward_all_workspace'] = (_sz, [_i64, _i32, _i32])
SIGNATURES['gs_sage3_backward_all'] = (_i32, [_c.POINTER(Sage3Bwd), _i64, _i32, _i32, _i32, _i32, _i32, _vp, _sz,
                                              _c.POINTER(AdamUpdate), _vp])
SIGNATURES['gs_sage2_backward_all_supported'] = (_i32, [_i32] * 4)
SIGNATURES['gs_sage2_backward_all_workspace'] = (_sz, [_i64, _i32, _i32, _i32])
SIGNATURES['gs_sage2_backward_all'] = (_i32, [_c.POINTER(Sage2BwdAll), _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                              _vp, _sz, _c.POINTER(AdamUpdate), _vp])
SIGNATURES['gs_sage2_supported'] = (_i32, [_i32] * 6)
SIGNATURES['gs_sage2_subtrees_supported'] = (_i32, [_i32] * 6)
SIGNATURES['gs_sage2_forward'] = (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i32,
                                         _u64, _u64, _vp, _vp,
                                         _c.POINTER(SageWeights), _c.POINTER(SageWeights),
                                         _c.POINTER(SageWeights), _c.POINTER(SageWeights),
                                         _i32, _i32, _i32, _vp, _vp, _vp, _c.POINTER(Sage2Saved), _vp, _vp])
SIGNATURES['gs_sage2_forward_subtrees'] = (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i32,
                                                  _u64, _u64, _vp, _vp,
                                                  _c.POINTER(SageWeights), _c.POINTER(SageWeights),
                                                  _c.POINTER(SageWeights), _c.POINTER(SageWeights),
                                                  _i32, _i32, _vp, _vp, _vp, _c.POINTER(Sage2Subtrees), _vp, _vp])
SIGNATURES['gs_sage3_seeds_supported'] = (_i32, [_i32] * 6)
SIGNATURES['gs_sage3_seeds_forward'] = (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp, _vp] +
                                        [_c.POINTER(SageWeights)] * 6 + [_i32, _i32, _i32, _i32, _i32, _vp, _vp])
SIGNATURES['gs_sage3_seeds_forward_saving'] = (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp, _vp] +
                                               [_c.POINTER(SageWeights)] * 6 + [_i32, _i32, _i32, _i32, _i32, _vp,
                                                                                _c.POINTER(Sage3SeedsSaved), _vp])
SIGNATURES['gs_sage2_backward_tail'] = (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32,
                                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(Sage2L0Mask), _vp])

SIGNATURES['gs_sage_project_wgrad_by_id_supported'] = (_i32, [_i32, _i32, _i32])
SIGNATURES['gs_sage2_tail_head_workspace'] = (_sz, [_i64, _i32])
SIGNATURES['gs_sage2_tail_head_rows'] = (_i32, [_i64])
SIGNATURES['gs_sage2_backward_tail_head'] = (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32,
                                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                                    _c.POINTER(Sage2L0Mask), _vp])
SIGNATURES['gs_link_head_reduce'] = (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp])
_f32 = _c.c_float
SIGNATURES['gs_link_head_workspace'] = (_sz, [_i64, _i32])
SIGNATURES['gs_link_head_fwd'] = (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp])
SIGNATURES['gs_link_head_mse_fwd_bwd'] = (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                 _vp, _sz, _vp])
SIGNATURES['gs_link_head_mse_fwd_bwd_static'] = (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                        _vp])
SIGNATURES['gs_stage_batch'] = (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _f32, _vp, _vp])
SIGNATURES['gs_adam_step'] = (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _vp, _f32, _f32, _f32, _f32, _vp])

_lib = None


class GsageError(RuntimeError):
    pass


def load():
    """Opens libgsage.so (once) and declares every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsageError(
            "libgsage.so is not built (%s missing): run `make -C graphsage_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback" % LIB_PATH)
    import torch  # noqa: F401  (loads the HIP runtime the library must share)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != GS_OK:
        msg = load().gs_last_error()
        raise GsageError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ''))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise GsageError("no MI355X visible: the GraphSAGE hot path runs only as HIP kernels on the GPU")
    return torch.device('cuda', torch.cuda.current_device())


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream
