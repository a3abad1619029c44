"""Tensor-level wrappers over the C ABI.

torch tensors are only the carrier of device memory here: every function checks
device, dtype, contiguity and shapes on the host (a kernel that is handed a
shape it does not expect faults the GPU), then passes raw pointers and the
current HIP stream to ``libgsage.so``.  Nothing in this file computes.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT, check, load, ptr, stream_handle


def _dev(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch tensor" % name)
    if not t.is_cuda:
        raise _lib.GsageError("%s must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dimensions (got %d)" % (name, ndim, t.dim()))
    return t


class LaunchTimer:
    """Exact duration of one kernel launch (``gs_timer_*``): ``arm()`` right before
    the call whose main kernel is to be timed, ``elapsed_us()`` afterwards."""

    def __init__(self):
        self._lib = load()
        h = ctypes.c_void_p()
        check(self._lib.gs_timer_create(ctypes.byref(h)), 'gs_timer_create')
        self._h = h

    def arm(self):
        check(self._lib.gs_timer_arm(self._h), 'gs_timer_arm')

    def elapsed_us(self):
        us = ctypes.c_float()
        check(self._lib.gs_timer_elapsed_us(self._h, ctypes.byref(us)), 'gs_timer_elapsed_us')
        return float(us.value)

    def __del__(self):
        try:
            self._lib.gs_timer_destroy(self._h)
        except Exception:
            pass


# Prepared calls keep raw device addresses of the weights.  Code that moves parameters into other
# storage (FlatGradBucket re-points every parameter into one flat buffer) announces it here; a
# prepared call compares one integer per run and re-reads the addresses when it has changed.
_storage_epoch = 0


def parameters_moved():
    global _storage_epoch
    _storage_epoch += 1


def storage_epoch():
    return _storage_epoch


def hop_sizes(fanouts):
    out, p = [], 1
    for n in fanouts:
        p *= int(n)
        out.append(p)
    return out


def build_csr(a, b, n_nodes, symmetric=False, sort_chunk=0):
    """Edge columns (int32 GPU tensors, entry i = edge a[i] -> b[i]; symmetric:
    followed by the flipped copies) -> (indptr int64[n_nodes+1], indices int32),
    neighbors in entry order.  Raises ValueError for ids outside [0, n_nodes)."""
    lib = load()
    _dev(a, torch.int32, 'a', 1)
    _dev(b, torch.int32, 'b', 1)
    if a.numel() != b.numel():
        raise ValueError("edge columns must have equal length")
    n_nodes, n_pairs = int(n_nodes), a.numel()
    if not 0 <= n_nodes < 2 ** 31:
        raise ValueError("node count must fit int32")
    need = ctypes.c_size_t(0)
    check(lib.gs_csr_build_workspace(n_pairs, n_nodes, int(sort_chunk), ctypes.byref(need)), 'gs_csr_build_workspace')
    ws = torch.empty(max(need.value, 1), dtype=torch.uint8, device=a.device)
    indptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=a.device)
    indices = torch.empty(n_pairs * (2 if symmetric else 1), dtype=torch.int32, device=a.device)
    err = torch.zeros(1, dtype=torch.int32, device=a.device)
    check(lib.gs_csr_build(ptr(a), ptr(b), n_pairs, 1 if symmetric else 0, n_nodes, ptr(indptr), ptr(indices),
                           ptr(err), int(sort_chunk), ptr(ws), ws.numel(), stream_handle()), 'gs_csr_build')
    if int(err.item()):
        raise ValueError("edge endpoints must be node ids in [0, %d)" % n_nodes)
    return indptr, indices


def sample(indptr, indices, seeds, fanouts, rng_seed=0, walk_id_base=0, walk_ids=None, replay=None,
           err_flag=None):
    """Multi-hop sampling; returns (list of int32 [n_seeds, size_h] tensors, err_flag)."""
    lib = load()
    _dev(indptr, torch.int64, 'indptr', 1)
    _dev(indices, torch.int32, 'indices', 1)
    _dev(seeds, torch.int32, 'seeds', 1)
    fanouts = [int(f) for f in fanouts]
    if not 1 <= len(fanouts) <= _lib.GS_MAX_HOPS:
        raise ValueError("between 1 and %d hops are supported" % _lib.GS_MAX_HOPS)
    if any(f < 1 for f in fanouts):
        raise ValueError("fan-outs must be positive")
    n_nodes = indptr.numel() - 1
    n_seeds = seeds.numel()
    sizes = hop_sizes(fanouts)
    if walk_ids is not None:
        _dev(walk_ids, torch.int64, 'walk_ids', 1)
        if walk_ids.numel() != n_seeds:
            raise ValueError("walk_ids must have one entry per seed")
    if replay is not None:
        _dev(replay, torch.int32, 'replay', 1)       # uint32 words carried in an int32 tensor
        if replay.numel() != n_seeds * sum(sizes):
            raise ValueError("replay tape must hold %d words (got %d)" % (n_seeds * sum(sizes), replay.numel()))
    if err_flag is None:
        err_flag = torch.zeros(2, dtype=torch.int32, device=seeds.device)
    else:
        _dev(err_flag, torch.int32, 'err_flag', 1)
    hops = [torch.empty((n_seeds, s), dtype=torch.int32, device=seeds.device) for s in sizes]
    fan_arr = (ctypes.c_int32 * len(fanouts))(*fanouts)
    out_arr = (ctypes.c_void_p * len(hops))(*[h.data_ptr() for h in hops])
    check(lib.gs_sample(ptr(indptr), ptr(indices), n_nodes, ptr(seeds), n_seeds, fan_arr, len(fanouts),
                        int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(walk_id_base) & 0xFFFFFFFFFFFFFFFF,
                        ptr(walk_ids), ptr(replay), out_arr, ptr(err_flag), stream_handle()), 'gs_sample')
    return hops, err_flag


def sample_cyclic(indptr, indices, seeds, fanouts, cursor, state, rng_seed=0, perm_tape=None, err_flag=None,
                  perm_tape_len=None):
    """``gs_sample_cyclic``: the reference's cyclic without-replacement walk.  ``cursor`` (int64 [n_nodes])
    and ``state`` (int64 [2]) are updated in place; returns (per-hop int32 tensors, err_flag)."""
    lib = load()
    _dev(indptr, torch.int64, 'indptr', 1)
    _dev(indices, torch.int32, 'indices', 1)
    _dev(seeds, torch.int32, 'seeds', 1)
    _dev(cursor, torch.int64, 'cursor', 1)
    _dev(state, torch.int64, 'state', 1)
    fanouts = [int(f) for f in fanouts]
    if not 1 <= len(fanouts) <= _lib.GS_MAX_HOPS:
        raise ValueError("between 1 and %d hops are supported" % _lib.GS_MAX_HOPS)
    if any(f < 1 for f in fanouts):
        raise ValueError("fan-outs must be positive")
    n_nodes = indptr.numel() - 1
    if cursor.numel() != n_nodes or state.numel() != 2:
        raise ValueError("cursor must have one entry per node and state two entries")
    if perm_tape is not None:
        _dev(perm_tape, torch.int32, 'perm_tape', 1)
        if perm_tape_len is not None and not 0 <= int(perm_tape_len) <= perm_tape.numel():
            raise ValueError("perm_tape_len outside the tape")
        if perm_tape.numel() == 0:
            raise ValueError("an empty tape has no address: pass a one-element tensor and perm_tape_len=0")
    if err_flag is None:
        err_flag = torch.zeros(2, dtype=torch.int32, device=seeds.device)
    else:
        _dev(err_flag, torch.int32, 'err_flag', 1)
    n_seeds = seeds.numel()
    hops = [torch.empty((n_seeds, s), dtype=torch.int32, device=seeds.device) for s in hop_sizes(fanouts)]
    fan_arr = (ctypes.c_int32 * len(fanouts))(*fanouts)
    out_arr = (ctypes.c_void_p * len(hops))(*[h.data_ptr() for h in hops])
    check(lib.gs_sample_cyclic(ptr(indptr), ptr(indices), n_nodes, ptr(seeds), n_seeds, fan_arr, len(fanouts), ptr(cursor),
                               int(rng_seed) & 0xFFFFFFFFFFFFFFFF, ptr(state), ptr(perm_tape),
                               0 if perm_tape is None else (perm_tape.numel() if perm_tape_len is None else int(perm_tape_len)),
                               out_arr, ptr(err_flag), stream_handle()),
          'gs_sample_cyclic')
    return hops, err_flag


def gather_rows(feat, ids, id_offset=0, out=None):
    lib = load()
    _dev(feat, torch.float32, 'feat', 2)
    _dev(ids, torch.int32, 'ids')
    n, f = ids.numel(), feat.shape[1]
    if out is None:
        out = torch.empty((n, f), dtype=torch.float32, device=feat.device)
    else:
        _dev(out, torch.float32, 'out')
        if out.numel() != n * f:
            raise ValueError("out has %d elements, %d needed" % (out.numel(), n * f))
    check(lib.gs_gather_rows(ptr(feat), feat.shape[0], f, int(id_offset), ptr(ids), n, ptr(out),
                             stream_handle()), 'gs_gather_rows')
    return out


def gather_max(feat, child_ids, n_parents, n_child, id_offset=0, out=None):
    """Per-feature maximum over each parent's children; arguments as gather_mean."""
    return gather_mean(feat, child_ids, n_parents, n_child, id_offset, out, _entry='gs_gather_max_fwd')


def group_max_bwd(g_max, x, out_max, n_child, g_x=None):
    """Gradient of the dense group max w.r.t. the children x [n_parents*n_child, F]."""
    lib = load()
    _dev(g_max, torch.float32, 'g_max', 2)
    _dev(x, torch.float32, 'x', 2)
    _dev(out_max, torch.float32, 'out_max', 2)
    n_parents, f = g_max.shape
    if out_max.shape != g_max.shape or x.shape != (n_parents * n_child, f):
        raise ValueError("shapes: g_max %s, out_max %s, x %s for n_child %d"
                         % (tuple(g_max.shape), tuple(out_max.shape), tuple(x.shape), n_child))
    if g_x is None:
        g_x = torch.empty_like(x)
    else:
        _dev(g_x, torch.float32, 'g_x')
        if g_x.numel() != x.numel():
            raise ValueError("g_x has %d elements, %d needed" % (g_x.numel(), x.numel()))
    check(lib.gs_group_max_bwd(ptr(g_max), ptr(x), ptr(out_max), n_parents, int(n_child), f, ptr(g_x),
                               stream_handle()), 'gs_group_max_bwd')
    return g_x


def gather_mean(feat, child_ids, n_parents, n_child, id_offset=0, out=None, _entry='gs_gather_mean_fwd'):
    """child_ids None: dense groups (parent p owns rows p*n_child .. of feat)."""
    lib = load()
    _dev(feat, torch.float32, 'feat', 2)
    f = feat.shape[1]
    if child_ids is not None:
        _dev(child_ids, torch.int32, 'child_ids')
        if child_ids.numel() != n_parents * n_child:
            raise ValueError("child_ids has %d entries, %d needed" % (child_ids.numel(), n_parents * n_child))
    elif n_parents * n_child > feat.shape[0]:
        raise ValueError("dense groups need n_parents*n_child <= rows of feat")
    if out is None:
        out = torch.empty((n_parents, f), dtype=torch.float32, device=feat.device)
    else:
        _dev(out, torch.float32, 'out')
        if out.numel() != n_parents * f:
            raise ValueError("out has %d elements, %d needed" % (out.numel(), n_parents * f))
    check(getattr(lib, _entry)(ptr(feat), feat.shape[0], f, int(id_offset), ptr(child_ids), n_parents,
                               n_child, ptr(out), stream_handle()), _entry)
    return out


def gather_mean_bwd(g_mean, child_ids, n_child, g_feat, id_offset=0):
    """Adds (child_ids given) or stores (dense) g_mean / n_child into g_feat rows."""
    lib = load()
    _dev(g_mean, torch.float32, 'g_mean', 2)
    _dev(g_feat, torch.float32, 'g_feat', 2)
    n_parents, f = g_mean.shape
    if g_feat.shape[1] != f:
        raise ValueError("feature widths differ")
    if child_ids is not None:
        _dev(child_ids, torch.int32, 'child_ids')
        if child_ids.numel() != n_parents * n_child:
            raise ValueError("child_ids has %d entries, %d needed" % (child_ids.numel(), n_parents * n_child))
    elif n_parents * n_child > g_feat.shape[0]:
        raise ValueError("dense groups need n_parents*n_child <= rows of g_feat")
    check(lib.gs_gather_mean_bwd(ptr(g_mean), ptr(child_ids), n_parents, n_child, f, int(id_offset),
                                 g_feat.shape[0], ptr(g_feat), stream_handle()), 'gs_gather_mean_bwd')
    return g_feat


def _project_args(x_self, x_mean, w_self, w_neigh, bias):
    _dev(x_self, torch.float32, 'x_self', 2)
    _dev(x_mean, torch.float32, 'x_mean', 2)
    _dev(w_self, torch.float32, 'w_self', 2)
    _dev(w_neigh, torch.float32, 'w_neigh', 2)
    m, fs = x_self.shape
    fn = x_mean.shape[1]
    if x_mean.shape[0] != m:
        raise ValueError("x_self and x_mean must have the same number of rows")
    hh = w_self.shape[1]
    if w_self.shape[0] != fs or w_neigh.shape != (fn, hh):
        raise ValueError("weight shapes %s / %s do not match inputs (%d, %d)" %
                         (tuple(w_self.shape), tuple(w_neigh.shape), fs, fn))
    if bias is not None:
        _dev(bias, torch.float32, 'bias', 1)
        if bias.numel() != 2 * hh:
            raise ValueError("bias must have %d entries" % (2 * hh))
    return m, fs, fn, 2 * hh


def sage_project(x_self, x_mean, w_self, w_neigh, bias, act, out=None):
    lib = load()
    m, fs, fn, h = _project_args(x_self, x_mean, w_self, w_neigh, bias)
    if out is None:
        out = torch.empty((m, h), dtype=torch.float32, device=x_self.device)
    else:
        _dev(out, torch.float32, 'out')
        if out.numel() != m * h:
            raise ValueError("out has %d elements, %d needed" % (out.numel(), m * h))
    check(lib.gs_sage_project_fwd(ptr(x_self), fs, ptr(x_mean), fn, m, ptr(w_self), ptr(w_neigh), ptr(bias),
                                  h, ACT[act], ptr(out), stream_handle()), 'gs_sage_project_fwd')
    return out


def sage_project_gather(feat_self, self_ids, id_offset, x_mean, w_self, w_neigh, bias, act, out=None):
    """sage_project with the own rows read from the feature table ``feat_self`` by
    node id (``self_ids`` int32[M], table row = id - id_offset) inside the kernel."""
    lib = load()
    _dev(feat_self, torch.float32, 'feat_self', 2)
    _dev(self_ids, torch.int32, 'self_ids', 1)
    _dev(x_mean, torch.float32, 'x_mean', 2)
    _dev(w_self, torch.float32, 'w_self', 2)
    _dev(w_neigh, torch.float32, 'w_neigh', 2)
    m, fs, fn, hh = self_ids.numel(), feat_self.shape[1], x_mean.shape[1], w_self.shape[1]
    if x_mean.shape[0] != m:
        raise ValueError("self_ids and x_mean must have the same number of rows")
    if w_self.shape[0] != fs or w_neigh.shape != (fn, hh):
        raise ValueError("weight shapes %s / %s do not match inputs (%d, %d)" %
                         (tuple(w_self.shape), tuple(w_neigh.shape), fs, fn))
    if bias is not None:
        _dev(bias, torch.float32, 'bias', 1)
        if bias.numel() != 2 * hh:
            raise ValueError("bias must have %d entries" % (2 * hh))
    h = 2 * hh
    if out is None:
        out = torch.empty((m, h), dtype=torch.float32, device=x_mean.device)
    else:
        _dev(out, torch.float32, 'out')
        if out.numel() != m * h:
            raise ValueError("out has %d elements, %d needed" % (out.numel(), m * h))
    check(lib.gs_sage_project_gather_fwd(ptr(feat_self), feat_self.shape[0], fs, int(id_offset), ptr(self_ids),
                                         ptr(x_mean), fn, m, ptr(w_self), ptr(w_neigh), ptr(bias), h, ACT[act],
                                         ptr(out), stream_handle()), 'gs_sage_project_gather_fwd')
    return out


def sage_project_bwd(x_self, x_mean, w_self, w_neigh, act, out, g_out, need_x_self=True, need_x_mean=True,
                     need_bias=True, g_w_self=None, g_w_neigh=None, g_bias=None, g_x_self=None, g_x_mean=None,
                     workspace=None):
    """Gradients of ``sage_project``.  Destination tensors may be passed in (views
    of a flat gradient bucket, preallocated activations); missing ones are
    allocated.  Returns (g_w_self, g_w_neigh, g_bias, g_x_self, g_x_mean)."""
    lib = load()
    m, fs, fn, h = _project_args(x_self, x_mean, w_self, w_neigh, None)
    _dev(g_out, torch.float32, 'g_out', 2)
    if tuple(g_out.shape) != (m, h):
        raise ValueError("g_out must be (%d, %d)" % (m, h))
    if out is not None:
        _dev(out, torch.float32, 'out', 2)
        if tuple(out.shape) != (m, h):
            raise ValueError("out must be (%d, %d)" % (m, h))
    dev = x_self.device

    def dest(t, shape, name, want=True):
        if t is None:
            return torch.empty(shape, dtype=torch.float32, device=dev) if want else None
        _dev(t, torch.float32, name)
        if t.numel() != int(torch.Size(shape).numel()):
            raise ValueError("%s has %d elements, %s needed" % (name, t.numel(), tuple(shape)))
        return t

    g_ws = dest(g_w_self, w_self.shape, 'g_w_self')
    g_wn = dest(g_w_neigh, w_neigh.shape, 'g_w_neigh')
    g_b = dest(g_bias, (h,), 'g_bias', need_bias)
    g_xs = dest(g_x_self, x_self.shape, 'g_x_self', need_x_self)
    g_xm = dest(g_x_mean, x_mean.shape, 'g_x_mean', need_x_mean)
    ws_bytes = lib.gs_sage_project_bwd_workspace(fs, fn, h, m)
    if workspace is None:
        workspace = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=dev)
    else:
        _dev(workspace, torch.float32, 'workspace', 1)
        if workspace.numel() * 4 < ws_bytes:
            raise ValueError("workspace of %d bytes needed" % ws_bytes)
    check(lib.gs_sage_project_bwd(ptr(x_self), fs, ptr(x_mean), fn, m, ptr(w_self), ptr(w_neigh), h, ACT[act],
                                  ptr(out), ptr(g_out), ptr(g_ws), ptr(g_wn), ptr(g_b), ptr(g_xs), ptr(g_xm),
                                  ptr(workspace), workspace.numel() * 4, stream_handle()), 'gs_sage_project_bwd')
    return g_ws, g_wn, g_b, g_xs, g_xm


def project_bwd_workspace_floats(fs, fn, h, m):
    return max(load().gs_sage_project_bwd_workspace(int(fs), int(fn), int(h), int(m)) // 4, 1)


def l2_normalize(x, out=None):
    lib = load()
    _dev(x, torch.float32, 'x', 2)
    if out is None:
        out = torch.empty_like(x)
    check(lib.gs_l2_normalize_fwd(ptr(x), x.shape[0], x.shape[1], ptr(out), stream_handle()),
          'gs_l2_normalize_fwd')
    return out


def l2_normalize_bwd(x, g_out, g_x=None):
    lib = load()
    _dev(x, torch.float32, 'x', 2)
    _dev(g_out, torch.float32, 'g_out', 2)
    if g_out.shape != x.shape:
        raise ValueError("shapes differ")
    if g_x is None:
        g_x = torch.empty_like(x)
    else:
        _dev(g_x, torch.float32, 'g_x', 2)
        if g_x.shape != x.shape:
            raise ValueError("shapes differ")
    check(lib.gs_l2_normalize_bwd(ptr(x), ptr(g_out), x.shape[0], x.shape[1], ptr(g_x), stream_handle()),
          'gs_l2_normalize_bwd')
    return g_x


class WgradPairCall:
    """A prepared ``gs_sage_project_wgrad_pair`` call: weight and bias gradients of
    two projections of one shape (the two node types of a layer), checked once,
    enqueued by ``run()``.  Each problem is a dict with x_self, x_mean, out (or
    None), g_out, g_w_self, g_w_neigh, g_bias (or None), workspace.  With
    'self_ids' (int32 [M]) and 'self_off' in a problem, its x_self is the feature
    TABLE and the own rows are read from it by id."""

    def __init__(self, problems, act):
        lib = load()
        if len(problems) != 2:
            raise ValueError("two problems expected")
        m, fs = problems[0]['x_mean'].shape[0], problems[0]['x_self'].shape[1]
        fn = problems[0]['x_mean'].shape[1]
        h = problems[0]['g_out'].shape[1]
        need = lib.gs_sage_project_bwd_workspace(fs, fn, h, m)
        self._keep, self._structs = [], []
        for p in problems:
            by_id = p.get('self_ids') is not None
            if by_id:
                _dev(p['self_ids'], torch.int32, 'self_ids', 1)
                if p['self_ids'].numel() != m:
                    raise ValueError("self_ids has %d entries, %d needed" % (p['self_ids'].numel(), m))
                if not lib.gs_sage_project_wgrad_by_id_supported(fs, fn, h):
                    raise _lib.GsageError("own rows by id are not available for Fs=%d Fn=%d H=%d" % (fs, fn, h))
            for name, shape in (('x_self', (m, fs)), ('x_mean', (m, fn)), ('g_out', (m, h)), ('out', (m, h))):
                if p.get(name) is None and name == 'out':
                    continue
                _dev(p[name], torch.float32, name, 2)
                if by_id and name == 'x_self':
                    if p[name].shape[1] != fs:
                        raise ValueError("the feature table must be %d wide" % fs)
                    continue
                if tuple(p[name].shape) != shape:
                    raise ValueError("%s must be %s (got %s)" % (name, shape, tuple(p[name].shape)))
            for name, n in (('g_w_self', fs * (h // 2)), ('g_w_neigh', fn * (h // 2)), ('g_bias', h)):
                if p.get(name) is None and name == 'g_bias':
                    continue
                _dev(p[name], torch.float32, name)
                if p[name].numel() != n:
                    raise ValueError("%s has %d elements, %d needed" % (name, p[name].numel(), n))
            _dev(p['workspace'], torch.float32, 'workspace', 1)
            if p['workspace'].numel() * 4 < need:
                raise ValueError("workspace of %d bytes needed" % need)
            if ACT[act] == ACT['relu'] and p.get('out') is None:
                raise ValueError("relu needs the forward output")
            self._keep.append(dict(p))
            self._structs.append(_lib.WgradProblem(ptr(p['x_self']), ptr(p['x_mean']), ptr(p.get('out')), ptr(p['g_out']),
                                                   ptr(p['g_w_self']), ptr(p['g_w_neigh']), ptr(p.get('g_bias')),
                                                   ptr(p['workspace']), p['workspace'].numel() * 4,
                                                   ptr(p.get('self_ids')), int(p.get('self_off', 0)),
                                                   p['x_self'].shape[0] if by_id else 0))
        self._args = (fs, fn, m, h, ACT[act])
        self._lib = lib
        self.epoch = _storage_epoch            # the gradient views it writes belong to this parameter layout

    def run(self):
        if self.epoch != _storage_epoch:
            raise _lib.GsageError("parameters were moved to other storage after this call was prepared; prepare it again")
        check(self._lib.gs_sage_project_wgrad_pair(ctypes.byref(self._structs[0]), ctypes.byref(self._structs[1]),
                                                   *self._args, stream_handle()), 'gs_sage_project_wgrad_pair')


def _l0_mask_struct(l0_mask, b, n0, h1):
    """gs_sage2_l0_mask from {'h1': {type: (>= B, H1) float32}, 'h1_mask': {type: int32 words}}, or None."""
    if l0_mask is None:
        return None, None
    st = _lib.Sage2L0Mask()
    st.input_scale = float(l0_mask.get('input_scale', 0.0))
    for i, t in enumerate(('user', 'item')):
        rows, words = l0_mask['h1'][t], l0_mask['h1_mask'][t]
        _dev(rows, torch.float32, 'l0_mask[h1][%s]' % t, 2)
        _dev(words, torch.int32, 'l0_mask[h1_mask][%s]' % t, 1)
        if rows.shape[0] < b or rows.shape[1] != h1:
            raise ValueError("l0_mask[h1][%s] must hold at least %d rows of %d" % (t, b, h1))
        if words.numel() != b * int(n0) * (h1 // 32):
            raise ValueError("l0_mask[h1_mask][%s] must have %d words" % (t, b * int(n0) * (h1 // 32)))
        st.h1[i] = rows.data_ptr()
        st.h1_mask[i] = words.data_ptr()
    return st, ctypes.byref(st)


def sage2_backward_tail(z, g_emb, w1_self, w1_neigh, n0, normalize, act, g_z, g_h1, l0_mask=None):
    """``gs_sage2_backward_tail``: every argument but the scalars is a dict by node
    type.  ``z`` / ``g_emb`` / ``g_z`` are (B, H2), ``w1_*`` (H1, H2/2), ``g_h1``
    (B * (1 + n0), H1) in the row layout of the one-kernel forward's saved tables.
    ``l0_mask``: the layer-0 relu masks (see ``_l0_mask_struct``); ``g_h1`` then leaves masked."""
    lib = load()
    types = ('user', 'item')
    b, h2 = z['user'].shape
    h1 = w1_self['user'].shape[0]
    for t in types:
        for name, d, shape in (('z', z, (b, h2)), ('g_emb', g_emb, (b, h2)), ('g_z', g_z, (b, h2)),
                               ('w1_self', w1_self, (h1, h2 // 2)), ('w1_neigh', w1_neigh, (h1, h2 // 2)),
                               ('g_h1', g_h1, (b * (1 + int(n0)), h1))):
            _dev(d[t], torch.float32, '%s[%s]' % (name, t), 2)
            if tuple(d[t].shape) != shape:
                raise ValueError("%s[%s] must be %s (got %s)" % (name, t, shape, tuple(d[t].shape)))
    check(lib.gs_sage2_backward_tail(ptr(z['user']), ptr(z['item']), ptr(g_emb['user']), ptr(g_emb['item']),
                                     b, int(n0), h1, h2, int(bool(normalize)), ACT[act],
                                     ptr(w1_self['user']), ptr(w1_neigh['user']), ptr(w1_self['item']), ptr(w1_neigh['item']),
                                     ptr(g_z['user']), ptr(g_z['item']), ptr(g_h1['user']), ptr(g_h1['item']),
                                     _l0_mask_struct(l0_mask, b, n0, h1)[1], stream_handle()), 'gs_sage2_backward_tail')


def sage2_backward_tail_head(z, emb, targets, head_kernel, head_bias, w1_self, w1_neigh, n0, normalize, act,
                             g_z, g_h1, pred, slab, l0_mask=None):
    """``gs_sage2_backward_tail_head``: ``sage2_backward_tail`` with the link head and its
    loss inside; ``emb`` = {'user','item'} embeddings (B, H2), ``slab`` the float32 workspace
    that ``link_head_reduce`` turns into the head's gradients and the loss."""
    lib = load()
    types = ('user', 'item')
    b, h2 = z['user'].shape
    h1 = w1_self['user'].shape[0]
    for t in types:
        for name, d, shape in (('z', z, (b, h2)), ('emb', emb, (b, h2)), ('g_z', g_z, (b, h2)),
                               ('w1_self', w1_self, (h1, h2 // 2)), ('w1_neigh', w1_neigh, (h1, h2 // 2)),
                               ('g_h1', g_h1, (b * (1 + int(n0)), h1))):
            _dev(d[t], torch.float32, '%s[%s]' % (name, t), 2)
            if tuple(d[t].shape) != shape:
                raise ValueError("%s[%s] must be %s (got %s)" % (name, t, shape, tuple(d[t].shape)))
    for name, t, n in (('targets', targets, b), ('pred', pred, b), ('head_kernel', head_kernel, 2 * h2), ('head_bias', head_bias, 1)):
        _dev(t, torch.float32, name)
        if t.numel() != n:
            raise ValueError("%s must have %d elements (got %d)" % (name, n, t.numel()))
    _dev(slab, torch.float32, 'slab', 1)
    check(lib.gs_sage2_backward_tail_head(ptr(z['user']), ptr(z['item']), ptr(emb['user']), ptr(emb['item']), ptr(targets),
                                          ptr(head_kernel), ptr(head_bias), b, int(n0), h1, h2, int(bool(normalize)), ACT[act],
                                          ptr(w1_self['user']), ptr(w1_neigh['user']), ptr(w1_self['item']), ptr(w1_neigh['item']),
                                          ptr(g_z['user']), ptr(g_z['item']), ptr(g_h1['user']), ptr(g_h1['item']),
                                          ptr(pred), ptr(slab), slab.numel() * 4,
                                          _l0_mask_struct(l0_mask, b, n0, h1)[1], stream_handle()),
          'gs_sage2_backward_tail_head')


class Sage2BackwardAll:
    """A prepared ``gs_sage2_backward_all`` call: the whole backward pass of the two-hop model on the saved
    tables of ``Sage2Call`` -- head and loss (when ``head`` is given), layer 1 backward, and every parameter
    gradient -- in one launch plus one reduction.  Dicts are keyed 'user' / 'item'.

    saved: z, h1, x_mean1, x_mean (layer-0 neighbor means), and either x_self (rows) or x_self = (table,
           first node id) with self_ids; h1_mask or None (layer 0 without activation)
    weights1 / grads: per type (w_self, w_neigh) of layer 1; grads[(layer, type)] = (g_w_self, g_w_neigh,
           g_bias or None)
    head: (emb dict, targets, kernel, bias, g_kernel, g_bias, pred, loss) or None, then ``g_emb`` dict at run()"""

    def __init__(self, saved, weights1, grads, n0, normalize, act, head=None, input_scale=1.0):
        lib = load()
        types = ('user', 'item')
        b, h2 = saved['z']['user'].shape
        h1 = weights1['user'][0].shape[0]
        by_id = saved.get('self_ids') is not None
        f = saved['x_mean']['user'].shape[1]
        rows = b * (1 + int(n0))
        if not lib.gs_sage2_backward_all_supported(f, saved['x_mean']['item'].shape[1], h1, h2):
            raise _lib.GsageError("gs_sage2_backward_all does not cover F=%d H1=%d H2=%d" % (f, h1, h2))
        st = _lib.Sage2BwdAll()
        self._keep = [saved, weights1, grads, head]
        for i, t in enumerate(types):
            for name, d, shape in (('z', saved['z'], (b, h2)), ('x_mean1', saved['x_mean1'], (b, h1)),
                                   ('x_mean', saved['x_mean'], (rows, f))):
                _dev(d[t], torch.float32, '%s[%s]' % (name, t), 2)
                if tuple(d[t].shape) != shape:
                    raise ValueError("%s[%s] must be %s (got %s)" % (name, t, shape, tuple(d[t].shape)))
            _dev(saved['h1'][t], torch.float32, 'h1[%s]' % t, 2)
            if saved['h1'][t].shape[0] < b or saved['h1'][t].shape[1] != h1:
                raise ValueError("h1[%s] must hold at least %d rows of %d" % (t, b, h1))
            st.z[i], st.x_mean1[i], st.x_mean[i], st.h1[i] = (ptr(saved['z'][t]), ptr(saved['x_mean1'][t]),
                                                              ptr(saved['x_mean'][t]), ptr(saved['h1'][t]))
            if by_id:
                table, first = saved['x_self'][t]
                _dev(table, torch.float32, 'feature table[%s]' % t, 2)
                _dev(saved['self_ids'][t], torch.int32, 'self_ids[%s]' % t, 1)
                if table.shape[1] != f or saved['self_ids'][t].numel() != rows:
                    raise ValueError("feature table[%s] must be %d wide, self_ids[%s] must have %d entries" % (t, f, t, rows))
                st.x_self[i], st.self_ids[i] = ptr(table), ptr(saved['self_ids'][t])
                st.self_off[i], st.self_rows[i] = int(first), table.shape[0]
            else:
                _dev(saved['x_self'][t], torch.float32, 'x_self[%s]' % t, 2)
                if tuple(saved['x_self'][t].shape) != (rows, f):
                    raise ValueError("x_self[%s] must be %s" % (t, (rows, f)))
                st.x_self[i] = ptr(saved['x_self'][t])
            if saved.get('h1_mask') is not None:
                _dev(saved['h1_mask'][t], torch.int32, 'h1_mask[%s]' % t, 1)
                if saved['h1_mask'][t].numel() != b * int(n0) * (h1 // 32):
                    raise ValueError("h1_mask[%s] must have %d words" % (t, b * int(n0) * (h1 // 32)))
                st.h1_mask[i] = ptr(saved['h1_mask'][t])
            ws, wn = weights1[t]
            for name, w in (('w1_self', ws), ('w1_neigh', wn)):
                _dev(w, torch.float32, '%s[%s]' % (name, t), 2)
                if tuple(w.shape) != (h1, h2 // 2):
                    raise ValueError("%s[%s] must be %s" % (name, t, (h1, h2 // 2)))
            st.w1_self[i], st.w1_neigh[i] = ptr(ws), ptr(wn)
            for layer, k, n, names in ((0, f, h1, ('g_w0_self', 'g_w0_neigh', 'g_b0')), (1, h1, h2, ('g_w1_self', 'g_w1_neigh', 'g_b1'))):
                g_ws, g_wn, g_b = grads[(layer, t)]
                for name, g, cnt in ((names[0], g_ws, k * (n // 2)), (names[1], g_wn, k * (n // 2)), (names[2], g_b, n)):
                    if g is None and name.startswith('g_b'):
                        continue
                    _dev(g, torch.float32, '%s[%s]' % (name, t))
                    if g.numel() != cnt:
                        raise ValueError("%s[%s] has %d elements, %d needed" % (name, t, g.numel(), cnt))
                    getattr(st, name)[i] = ptr(g)
        self._head = head is not None
        if head is not None:
            emb, targets, kernel, bias, g_kernel, g_bias, pred, loss = head
            for name, x, n in (('targets', targets, b), ('pred', pred, b), ('head_kernel', kernel, 2 * h2), ('head_bias', bias, 1),
                               ('g_kernel', g_kernel, 2 * h2), ('g_bias', g_bias, 1), ('loss', loss, 1)):
                _dev(x, torch.float32, name)
                if x.numel() != n:
                    raise ValueError("%s must have %d elements (got %d)" % (name, n, x.numel()))
            for i, t in enumerate(types):
                _dev(emb[t], torch.float32, 'emb[%s]' % t, 2)
                if tuple(emb[t].shape) != (b, h2):
                    raise ValueError("emb[%s] must be %s" % (t, (b, h2)))
                st.emb[i] = ptr(emb[t])
            st.targets, st.head_kernel, st.head_bias = ptr(targets), ptr(kernel), ptr(bias)
            st.g_head_kernel, st.g_head_bias, st.pred, st.loss = ptr(g_kernel), ptr(g_bias), ptr(pred), ptr(loss)
        st.input_scale = float(input_scale)
        need = lib.gs_sage2_backward_all_workspace(b, f, h1, h2)
        self._ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=saved['z']['user'].device)
        self._st, self._lib = st, lib
        self._args = (b, int(n0), f, h1, h2, int(bool(normalize)), ACT[act])
        self._shape = (b, h2)
        self.epoch = _storage_epoch            # the gradient views it writes belong to this parameter layout

    def run(self, g_emb=None, targets=None, adam=None):
        """adam: None, or a dict for the optimizer step in the same reduction: grads, params, m, v (flat float32
        buffers; every gradient this call writes must be a view into ``grads``), b1, b2, eps, and the rate as
        'lr' (a float) or 'lr_dev' (a one-element tensor, for captured graphs); optional grad_scale."""
        if self.epoch != _storage_epoch:
            raise _lib.GsageError("parameters were moved to other storage after this call was prepared; prepare it again")
        if self._head:
            if targets is not None:
                _dev(targets, torch.float32, 'targets')
                if targets.numel() != self._shape[0]:
                    raise ValueError("targets must have %d elements" % self._shape[0])
                self._st.targets = ptr(targets)
                self._keep_targets = targets
        else:
            for i, t in enumerate(('user', 'item')):
                _dev(g_emb[t], torch.float32, 'g_emb[%s]' % t, 2)
                if tuple(g_emb[t].shape) != self._shape:
                    raise ValueError("g_emb[%s] must be %s" % (t, self._shape))
                self._st.g_emb[i] = ptr(g_emb[t])
            self._keep_g = g_emb
        upd = None
        if adam is not None:
            for name in ('grads', 'params', 'm', 'v'):
                _dev(adam[name], torch.float32, 'adam[%s]' % name, 1)
                if adam[name].numel() != adam['grads'].numel():
                    raise ValueError("adam[%s] must have %d elements" % (name, adam['grads'].numel()))
            lr_dev = adam.get('lr_dev')
            if lr_dev is not None:
                _dev(lr_dev, torch.float32, 'adam[lr_dev]')
            self._adam = _lib.AdamUpdate(ptr(adam['grads']), ptr(adam['params']), ptr(adam['m']), ptr(adam['v']),
                                         adam['grads'].numel(), ptr(lr_dev), float(adam.get('lr', 0.0)), float(adam['b1']),
                                         float(adam['b2']), float(adam['eps']), float(adam.get('grad_scale', 1.0)))
            self._keep_adam = adam
            upd = ctypes.byref(self._adam)
        check(self._lib.gs_sage2_backward_all(ctypes.byref(self._st), *self._args, ptr(self._ws), self._ws.numel() * 4,
                                              upd, stream_handle()), 'gs_sage2_backward_all')


def _adam_struct(adam):
    """gs_adam_update from the dict of ``Sage2BackwardAll.run``; returns (struct or None, what to keep alive)."""
    if adam is None:
        return None
    for name in ('grads', 'params', 'm', 'v'):
        _dev(adam[name], torch.float32, 'adam[%s]' % name, 1)
        if adam[name].numel() != adam['grads'].numel():
            raise ValueError("adam[%s] must have %d elements" % (name, adam['grads'].numel()))
    lr_dev = adam.get('lr_dev')
    if lr_dev is not None:
        _dev(lr_dev, torch.float32, 'adam[lr_dev]')
    return _lib.AdamUpdate(ptr(adam['grads']), ptr(adam['params']), ptr(adam['m']), ptr(adam['v']),
                           adam['grads'].numel(), ptr(lr_dev), float(adam.get('lr', 0.0)), float(adam['b1']),
                           float(adam['b2']), float(adam['eps']), float(adam.get('grad_scale', 1.0)))


class Sage3BackwardAll:
    """A prepared ``gs_sage3_backward_all`` call: the whole backward pass of the three-hop model (32 units per layer)
    on the layered forward's tables -- two kernels and one reduction.  Dicts keyed 'user' / 'item'.

    saved: z [B, 32], h0, x_mean0, row_ids (layer-0 tables, B (1 + n0 + n0 n1) rows), h1, x_mean1 (layer-1 tables,
           B (1 + n0) rows), x_mean2 [B, 32], feat = (table, first node id) per type
    weights: {1: {type: (w_self, w_neigh)}, 2: {...}};  grads[(layer, type)] = (g_w_self, g_w_neigh, g_bias or None)
    head: as ``Sage2BackwardAll``; fan-outs n0 (hop 1), n1 (hop 2); act: layer 2's activation
    saved['hop1'] = {'h0', 'h1', 'x_mean0', 'x_mean1', 'ids'} and saved['hop2'] = {'mask' (or 'h0'), 'x_mean0', 'ids'} (each ->
    {type: tensor}): the rows below the seeds live apart (what the subtree and seeds kernels' saving forms leave); the
    tables above then hold the seeds' B rows only"""

    def __init__(self, saved, weights, grads, n0, n1, normalize, act, head=None):
        lib = load()
        types = ('user', 'item')
        b = saved['z']['user'].shape[0]
        f = saved['x_mean0']['user'].shape[1]
        rows0, rows1 = b * (1 + n0 + n0 * n1), b * (1 + n0)
        split = saved.get('hop1') is not None
        if split:
            rows0 = rows1 = b
        if not lib.gs_sage3_backward_all_supported(f, saved['x_mean0']['item'].shape[1], weights[1]['user'][0].shape[0],
                                                   2 * weights[1]['user'][0].shape[1], saved['z']['user'].shape[1], int(n0), int(n1)):
            raise _lib.GsageError("gs_sage3_backward_all does not cover this shape")
        st = _lib.Sage3Bwd()
        self._keep = [saved, weights, grads, head]
        for i, t in enumerate(types):
            for name, shape in (('z', (b, 32)), ('x_mean2', (b, 32)), ('h1', (rows1, 32)), ('x_mean1', (rows1, 32)),
                                ('h0', (rows0, 32)), ('x_mean0', (rows0, f))):
                x = saved[name][t]
                _dev(x, torch.float32, '%s[%s]' % (name, t), 2)
                if tuple(x.shape) != shape:
                    raise ValueError("%s[%s] must be %s (got %s)" % (name, t, shape, tuple(x.shape)))
                getattr(st, name)[i] = ptr(x)
            _dev(saved['row_ids'][t], torch.int32, 'row_ids[%s]' % t, 1)
            if saved['row_ids'][t].numel() != rows0:
                raise ValueError("row_ids[%s] must have %d entries" % (t, rows0))
            st.row_ids[i] = ptr(saved['row_ids'][t])
            table, first = saved['feat'][t]
            _dev(table, torch.float32, 'feature table[%s]' % t, 2)
            if table.shape[1] != f:
                raise ValueError("feature table[%s] must be %d wide" % (t, f))
            st.feat[i], st.feat_off[i], st.feat_rows[i] = ptr(table), int(first), table.shape[0]
            for layer, names in ((1, ('w1_self', 'w1_neigh')), (2, ('w2_self', 'w2_neigh'))):
                for name, w in zip(names, weights[layer][t]):
                    _dev(w, torch.float32, '%s[%s]' % (name, t), 2)
                    if tuple(w.shape) != (32, 16):
                        raise ValueError("%s[%s] must be (32, 16)" % (name, t))
                    getattr(st, name)[i] = ptr(w)
            for layer, k in ((0, f), (1, 32), (2, 32)):
                g_ws, g_wn, g_b = grads[(layer, t)]
                for name, g, cnt in (('g_w%d_self' % layer, g_ws, k * 16), ('g_w%d_neigh' % layer, g_wn, k * 16), ('g_b%d' % layer, g_b, 32)):
                    if g is None and name.startswith('g_b'):
                        continue
                    _dev(g, torch.float32, '%s[%s]' % (name, t))
                    if g.numel() != cnt:
                        raise ValueError("%s[%s] has %d elements, %d needed" % (name, t, g.numel(), cnt))
                    getattr(st, name)[i] = ptr(g)
        if split:
            h1s, h2s = _lib.Sage3BwdHop1(), _lib.Sage3BwdHop2()
            nh1, nh2 = b * n0, b * n0 * n1
            for i, t in enumerate(types):
                for name, shape, dt in (('h0', (nh1, 32), torch.float32), ('h1', (nh1, 32), torch.float32),
                                        ('x_mean0', (nh1, f), torch.float32), ('x_mean1', (nh1, 32), torch.float32),
                                        ('ids', (nh1,), torch.int32)):
                    x = saved['hop1'][name][t]
                    _dev(x, dt, 'hop1[%s][%s]' % (name, t), len(shape))
                    if tuple(x.shape) != shape:
                        raise ValueError("hop1[%s][%s] must be %s (got %s)" % (name, t, shape, tuple(x.shape)))
                    getattr(h1s, name)[i] = ptr(x)
                for name, shape, dt in (('mask', (nh2,), torch.int32), ('h0', (nh2, 32), torch.float32),
                                        ('x_mean0', (nh2, f), torch.float32), ('ids', (nh2,), torch.int32)):
                    x = (saved['hop2'].get(name) or {}).get(t)
                    if x is None and name in ('mask', 'h0'):
                        continue
                    _dev(x, dt, 'hop2[%s][%s]' % (name, t), len(shape))
                    if tuple(x.shape) != shape:
                        raise ValueError("hop2[%s][%s] must be %s (got %s)" % (name, t, shape, tuple(x.shape)))
                    getattr(h2s, name)[i] = ptr(x)
            self._hops = (h1s, h2s)
            st.hop1, st.hop2 = ctypes.pointer(h1s), ctypes.pointer(h2s)
        self._head = head is not None
        if head is not None:
            emb, targets, kernel, bias, g_kernel, g_bias, pred, loss = head
            for name, x, n in (('targets', targets, b), ('pred', pred, b), ('head_kernel', kernel, 64), ('head_bias', bias, 1),
                               ('g_kernel', g_kernel, 64), ('g_bias', g_bias, 1), ('loss', loss, 1)):
                _dev(x, torch.float32, name)
                if x.numel() != n:
                    raise ValueError("%s must have %d elements (got %d)" % (name, n, x.numel()))
            for i, t in enumerate(types):
                _dev(emb[t], torch.float32, 'emb[%s]' % t, 2)
                if tuple(emb[t].shape) != (b, 32):
                    raise ValueError("emb[%s] must be %s" % (t, (b, 32)))
                st.emb[i] = ptr(emb[t])
            st.targets, st.head_kernel, st.head_bias = ptr(targets), ptr(kernel), ptr(bias)
            st.g_head_kernel, st.g_head_bias, st.pred, st.loss = ptr(g_kernel), ptr(g_bias), ptr(pred), ptr(loss)
        need = lib.gs_sage3_backward_all_workspace(b, int(n0), f)
        self._ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=saved['z']['user'].device)
        self._st, self._lib = st, lib
        self._args = (b, int(n0), int(n1), f, int(bool(normalize)), ACT[act])
        self._b = b
        self.epoch = _storage_epoch

    def run(self, g_emb=None, targets=None, adam=None):
        if self.epoch != _storage_epoch:
            raise _lib.GsageError("parameters were moved to other storage after this call was prepared; prepare it again")
        if self._head:
            if targets is not None:
                _dev(targets, torch.float32, 'targets')
                if targets.numel() != self._b:
                    raise ValueError("targets must have %d elements" % self._b)
                self._st.targets = ptr(targets)
                self._keep_targets = targets
        else:
            for i, t in enumerate(('user', 'item')):
                _dev(g_emb[t], torch.float32, 'g_emb[%s]' % t, 2)
                if tuple(g_emb[t].shape) != (self._b, 32):
                    raise ValueError("g_emb[%s] must be %s" % (t, (self._b, 32)))
                self._st.g_emb[i] = ptr(g_emb[t])
            self._keep_g = g_emb
        self._adam = _adam_struct(adam)
        self._keep_adam = adam
        check(self._lib.gs_sage3_backward_all(ctypes.byref(self._st), *self._args, ptr(self._ws), self._ws.numel() * 4,
                                              None if self._adam is None else ctypes.byref(self._adam), stream_handle()),
              'gs_sage3_backward_all')


def sage2_backward_all_supported(fu, fi, h1, h2):
    return bool(load().gs_sage2_backward_all_supported(int(fu), int(fi), int(h1), int(h2)))


def tail_head_slab(b, h2, device):
    """Workspace of ``sage2_backward_tail_head`` for B pairs and H2-wide embeddings."""
    n = load().gs_sage2_tail_head_workspace(int(b), int(h2))
    return torch.empty(max(n // 4, 1), dtype=torch.float32, device=device)


def link_head_reduce(slab, b, e, g_kernel, g_bias, loss):
    """Adds the slab rows of ``sage2_backward_tail_head`` into the head's gradients and the loss."""
    lib = load()
    _dev(slab, torch.float32, 'slab', 1)
    for name, t, n in (('g_kernel', g_kernel, 2 * e), ('g_bias', g_bias, 1), ('loss', loss, 1)):
        _dev(t, torch.float32, name)
        if t.numel() != n:
            raise ValueError("%s must have %d elements" % (name, n))
    rows = lib.gs_sage2_tail_head_rows(int(b))
    if slab.numel() < rows * (2 * e + 2):
        raise ValueError("slab too small")
    check(lib.gs_link_head_reduce(ptr(slab), rows, int(e), int(b), ptr(g_kernel), ptr(g_bias), ptr(loss), stream_handle()),
          'gs_link_head_reduce')


def sage2_backward_tail_supported(h1, h2):
    return h1 in (32, 64) and h2 in (32, 64)


def fused_supported(fs, fn, h, max_child):
    return bool(load().gs_sage_fused_supported(int(fs), int(fn), int(h), int(max_child)))


class FusedLayerCall:
    """A prepared ``gs_sage_fused_fwd`` / ``gs_sage_fused_sampled_fwd`` call: segment
    table and argument checks are done once, ``run()`` only enqueues the kernel.

    ``sampling`` = {'indptr', 'indices', 'err_flag', 'tape_len'} switches to the
    entry point that draws children inside the kernel; a segment that is to be
    drawn carries 'draw' = {'hop', 'heads_per_walk', 'tape_off', 'walk0',
    'out_child_ids' (optional)} instead of 'child_ids'.  The key, walk ids and tape
    of a step are arguments of ``run``."""

    def __init__(self, feat_self, off_self, feat_neigh, off_neigh, segments, w_self, w_neigh, bias, act,
                 normalize, out, save_self=None, save_mean=None, sampling=None, exact_pieces=None):
        """``exact_pieces``: wide layers only.  True / None (default): the call owns a workspace and the product
        runs on bf16 pieces of the fp32 operands (``gs_sage_fused_fwd_ws``); False: the fp32 matrix instruction.
        The environment variable GSAGE_WIDE_F32=1 makes False the default."""
        lib = load()
        _dev(feat_self, torch.float32, 'feat_self', 2)
        _dev(feat_neigh, torch.float32, 'feat_neigh', 2)
        _dev(w_self, torch.float32, 'w_self', 2)
        _dev(w_neigh, torch.float32, 'w_neigh', 2)
        _dev(out, torch.float32, 'out', 2)
        fs, fn, hh = feat_self.shape[1], feat_neigh.shape[1], w_self.shape[1]
        if tuple(w_self.shape) != (fs, hh) or tuple(w_neigh.shape) != (fn, hh):
            raise ValueError("weight shapes do not match the feature widths")
        if bias is not None:
            _dev(bias, torch.float32, 'bias', 1)
            if bias.numel() != 2 * hh:
                raise ValueError("bias must have %d entries" % (2 * hh))
        rows = 0
        self._keep = []
        arr = (_lib.FusedSegment * len(segments))()
        for i, sg in enumerate(segments):
            n_heads, n_child = int(sg['n_heads']), int(sg['n_child'])
            head_ids, child_ids = sg.get('head_ids'), sg.get('child_ids')
            if head_ids is not None:
                _dev(head_ids, torch.int32, 'head_ids')
                if head_ids.numel() != n_heads:
                    raise ValueError("segment %d: head_ids has %d entries, %d needed" % (i, head_ids.numel(), n_heads))
            if child_ids is not None:
                _dev(child_ids, torch.int32, 'child_ids')
                if child_ids.numel() != n_heads * n_child:
                    raise ValueError("segment %d: child_ids has %d entries, %d needed" %
                                     (i, child_ids.numel(), n_heads * n_child))
            if sg.get('draw') is not None:
                if sampling is None or head_ids is None or child_ids is not None:
                    raise ValueError("segment %d: a drawn segment needs `sampling`, head ids and no child ids" % i)
                ids_out = sg['draw'].get('out_child_ids')
                if ids_out is not None:
                    _dev(ids_out, torch.int32, 'out_child_ids')
                    if ids_out.numel() != n_heads * n_child:
                        raise ValueError("segment %d: out_child_ids has %d entries, %d needed" %
                                         (i, ids_out.numel(), n_heads * n_child))
                    self._keep.append(ids_out)
            arr[i] = _lib.FusedSegment(ptr(head_ids), ptr(child_ids), n_heads, int(sg.get('self_row0', 0)),
                                       int(sg.get('child_row0', 0)), n_child)
            self._keep += [head_ids, child_ids]
            rows += n_heads
        if tuple(out.shape) != (rows, 2 * hh):
            raise ValueError("out must be (%d, %d)" % (rows, 2 * hh))
        for name, t, width in (('save_self', save_self, fs), ('save_mean', save_mean, fn)):
            if t is not None:
                _dev(t, torch.float32, name, 2)
                if tuple(t.shape) != (rows, width):
                    raise ValueError("%s must be (%d, %d)" % (name, rows, width))
        max_child = max([int(sg['n_child']) for sg in segments] or [1])
        if not lib.gs_sage_fused_supported(fs, fn, 2 * hh, max_child):
            raise _lib.GsageError("fused layer kernel does not cover Fs=%d Fn=%d H=%d n_child=%d" %
                                  (fs, fn, 2 * hh, max_child))
        self._lib = lib
        self._arr = arr
        self._live_heads = {}
        self._keep += [feat_self, feat_neigh, w_self, w_neigh, bias, out, save_self, save_mean]
        self._weights = (w_self, w_neigh, bias)
        self._sampling = None
        if sampling is not None:
            for name, dtype in (('indptr', torch.int64), ('indices', torch.int32), ('err_flag', torch.int32)):
                _dev(sampling[name], dtype, name, 1)
            self._draws = (_lib.FusedDraw * len(segments))()
            self._walks = 0
            for i, sg in enumerate(segments):
                d = sg.get('draw')
                if d is None:
                    self._draws[i] = _lib.FusedDraw(-1, 1, 0, 0, None)
                    continue
                hpw, walk0 = int(d['heads_per_walk']), int(d['walk0'])
                if int(sg['n_heads']) % hpw:
                    raise ValueError("segment %d: %d heads are not whole walks of %d" % (i, int(sg['n_heads']), hpw))
                self._walks = max(self._walks, walk0 + int(sg['n_heads']) // hpw)
                self._draws[i] = _lib.FusedDraw(int(d['hop']), hpw, int(d['tape_off']), walk0, ptr(d.get('out_child_ids')))
            self._sampling = sampling
            self._tape_len = int(sampling['tape_len'])
            self._keep += [sampling['indptr'], sampling['indices'], sampling['err_flag']]
        if exact_pieces is None:
            exact_pieces = not os.environ.get('GSAGE_WIDE_F32')
        self._workspace = None
        ws_bytes = lib.gs_sage_fused_workspace_bytes(fs, fn, 2 * hh, max_child) if exact_pieces and sampling is None else 0
        if ws_bytes:
            self._workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=feat_self.device)
        self._head = (ptr(feat_self), feat_self.shape[0], fs, int(off_self), ptr(feat_neigh), feat_neigh.shape[0],
                      fn, int(off_neigh), arr, len(segments))
        self._tail = (2 * hh, ACT[act], int(bool(normalize)), ptr(out), ptr(save_self), ptr(save_mean))
        self._read_weight_addresses()
        self.out = out

    def _read_weight_addresses(self):
        self._args = self._head + tuple(ptr(w) for w in self._weights) + self._tail
        self._epoch = _storage_epoch

    def set_head_ids(self, seg, head_ids):
        """Points segment ``seg`` at another head-id array of the same length."""
        _dev(head_ids, torch.int32, 'head_ids')
        if head_ids.numel() != self._arr[seg].n_heads:
            raise ValueError("head_ids has %d entries, %d needed" % (head_ids.numel(), self._arr[seg].n_heads))
        self._arr[seg].head_ids = head_ids.data_ptr()
        self._live_heads[seg] = head_ids

    def run(self, rng_seed=0, walk_id_base=0, walk_ids=None, replay=None):
        if self._epoch != _storage_epoch:
            self._read_weight_addresses()
        if self._sampling is None:
            if self._workspace is not None:
                check(self._lib.gs_sage_fused_fwd_ws(*self._args, self._workspace.data_ptr(), self._workspace.numel(),
                                                     stream_handle()), 'gs_sage_fused_fwd_ws')
            else:
                check(self._lib.gs_sage_fused_fwd(*self._args, stream_handle()), 'gs_sage_fused_fwd')
            return self.out
        sp = self._sampling
        if walk_ids is not None:
            _dev(walk_ids, torch.int64, 'walk_ids', 1)
            if walk_ids.numel() < self._walks:
                raise ValueError("walk_ids has %d entries, the drawn segments span %d walks" % (walk_ids.numel(), self._walks))
        if replay is not None:
            _dev(replay, torch.int32, 'replay', 1)
            if replay.numel() < self._walks * self._tape_len:
                raise ValueError("replay tape has %d words, %d needed" % (replay.numel(), self._walks * self._tape_len))
        desc = _lib.FusedSampling(ptr(sp['indptr']), ptr(sp['indices']), sp['indptr'].numel() - 1,
                                  int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(walk_id_base) & 0xFFFFFFFFFFFFFFFF,
                                  ptr(walk_ids), ptr(replay), self._tape_len, ptr(sp['err_flag']))
        check(self._lib.gs_sage_fused_sampled_fwd(*self._args, ctypes.byref(desc), self._draws, stream_handle()),
              'gs_sage_fused_sampled_fwd')
        return self.out


def sage_fused(feat_self, off_self, feat_neigh, off_neigh, segments, w_self, w_neigh, bias, act,
               normalize=False, out=None, save_self=None, save_mean=None, exact_pieces=None):
    """One aggregator layer in one kernel; see include/gsage.h: gs_sage_fused_fwd / gs_sage_fused_fwd_ws."""
    rows = sum(int(sg['n_heads']) for sg in segments)
    if out is None:
        out = torch.empty((rows, 2 * w_self.shape[1]), dtype=torch.float32, device=feat_self.device)
    return FusedLayerCall(feat_self, off_self, feat_neigh, off_neigh, segments, w_self, w_neigh, bias, act,
                          normalize, out, save_self, save_mean, exact_pieces=exact_pieces).run()


def dropout_seed_of(rng_seed):
    """Key of a step's dropout masks, from the key of its sampling draws (another stream of the same generator)."""
    return (int(rng_seed) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def sage2_supported(fu, fi, h1, h2, n0, n1):
    return bool(load().gs_sage2_supported(int(fu), int(fi), int(h1), int(h2), int(n0), int(n1)))


class Sage2Call:
    """A prepared ``gs_sage2_forward`` call (whole two-hop forward, one kernel).

    ``weights`` = {(layer, node_type): (w_self, w_neigh, bias or None)} for layers
    0, 1 and node types 'user', 'item'.
    """

    def __init__(self, indptr, indices, feat_user, feat_item, fanouts, weights, normalize, n_pairs,
                 emb=None, keep_ids=False, saved=None, subtrees=None, hop_buffers=None):
        """``saved`` (training): {'x_self', 'x_mean', 'h1', 'x_mean1', 'z'} -> {'user': tensor, 'item':
        tensor}, the buffers the kernel fills for the backward pass (layout in include/gsage.h).

        ``subtrees`` = {'roots_per_walk': R, 'tape_len', 'tape_off_hop2', 'tape_off_hop3', 'out_h0'
        (tensor (2*n_pairs*R, H1) or None)}: ``gs_sage2_forward_subtrees`` -- the roots given to
        ``run`` are the 2*n_pairs*R hop-1 nodes of a three-hop model's walks, ``fanouts`` those of
        hops 2 and 3, ``emb`` (2*n_pairs*R, H2) receives relu(layer 1) at every root.  With
        'draw_hop1': int32 tensor (2*n_pairs*R) the kernel draws hop 1 too: ``run`` then takes the
        2*n_pairs seed nodes and the hop-1 ids are written to that tensor.
        ``hop_buffers`` = (int32 tensor or None, int32 tensor or None): where the drawn ids go (instead of ``keep_ids``).
        subtrees['saved'] (training) = {'x_mean0_hop2', 'mask_hop2', 'x_mean0_hop1', 'x_mean1_hop1'} -> {'user': tensor,
        'item': tensor}: the tables of the rows below the first hop that ``Sage3BackwardAll`` reads (include/gsage.h)."""
        lib = load()
        _dev(indptr, torch.int64, 'indptr', 1)
        _dev(indices, torch.int32, 'indices', 1)
        _dev(feat_user, torch.float32, 'feat_user', 2)
        _dev(feat_item, torch.float32, 'feat_item', 2)
        if len(fanouts) != 2:
            raise ValueError("the one-kernel forward covers exactly two hops")
        n0, n1 = int(fanouts[0]), int(fanouts[1])
        nu, fu = feat_user.shape
        ni, fi = feat_item.shape
        if indptr.numel() != nu + ni + 1:
            raise ValueError("indptr must have n_users + n_items + 1 entries")
        self._w = []
        self._weights = weights
        self._epoch = _storage_epoch
        dims = {}
        for layer in (0, 1):
            for t in ('user', 'item'):
                w_self, w_neigh, bias = weights[(layer, t)]
                _dev(w_self, torch.float32, 'w_self', 2)
                _dev(w_neigh, torch.float32, 'w_neigh', 2)
                if bias is not None:
                    _dev(bias, torch.float32, 'bias', 1)
                dims[(layer, t)] = (w_self.shape[0], w_neigh.shape[0], 2 * w_self.shape[1])
                if w_neigh.shape[1] != w_self.shape[1] or (bias is not None and bias.numel() != 2 * w_self.shape[1]):
                    raise ValueError("inconsistent weight shapes for layer %d, %s" % (layer, t))
                self._w.append(_lib.SageWeights(ptr(w_self), ptr(w_neigh), ptr(bias)))
        h1, h2 = dims[(0, 'user')][2], dims[(1, 'user')][2]
        want = {(0, 'user'): (fu, fi, h1), (0, 'item'): (fi, fu, h1), (1, 'user'): (h1, h1, h2), (1, 'item'): (h1, h1, h2)}
        if dims != want:
            raise ValueError("weight shapes %s do not fit the model %s" % (dims, want))
        query = lib.gs_sage2_subtrees_supported if subtrees is not None else lib.gs_sage2_supported
        if not query(fu, fi, h1, h2, n0, n1):
            raise _lib.GsageError("the one-kernel forward does not cover Fu=%d Fi=%d H=[%d,%d] fan-outs [%d,%d]" %
                                  (fu, fi, h1, h2, n0, n1))
        dev = indptr.device
        self.n_pairs = int(n_pairs)
        self._sub = None
        self.n_roots = 2 * self.n_pairs
        if subtrees is not None:
            if saved is not None:
                raise ValueError("the subtree form has no saved-activation variant")
            r = int(subtrees['roots_per_walk'])
            self.n_roots = 2 * self.n_pairs * r
            out_h0 = subtrees.get('out_h0')
            if out_h0 is not None:
                _dev(out_h0, torch.float32, 'out_h0', 2)
                if tuple(out_h0.shape) != (self.n_roots, h1):
                    raise ValueError("out_h0 must be (%d, %d)" % (self.n_roots, h1))
            hop1_out = subtrees.get('draw_hop1')
            if hop1_out is not None:
                _dev(hop1_out, torch.int32, 'draw_hop1')
                if hop1_out.numel() != self.n_roots:
                    raise ValueError("draw_hop1 must have %d entries" % self.n_roots)
            self._sub = _lib.Sage2Subtrees(r, int(subtrees.get('tape_len', 0)), int(subtrees.get('tape_off_hop2', 0)),
                                           int(subtrees.get('tape_off_hop3', 0)), ptr(out_h0), None, ptr(hop1_out))
            self._sub_keep = (out_h0, hop1_out, subtrees.get('saved'))
            sv = subtrees.get('saved')
            if sv is not None:
                # training: the tables gs_sage3_backward_all reads below the first hop, by the node type of the rows
                per_type = self.n_pairs * r
                for i, t in enumerate(('user', 'item')):
                    fo = fi if t == 'user' else fu
                    for name, shape, dt in (('x_mean0_hop2', (per_type * n0, fo), torch.float32),
                                            ('mask_hop2', (per_type * n0,), torch.int32),
                                            ('x_mean0_hop1', (per_type, fo), torch.float32),
                                            ('x_mean1_hop1', (per_type, h1), torch.float32)):
                        buf = sv[name][t]
                        _dev(buf, dt, 'saved[%s][%s]' % (name, t), len(shape))
                        if tuple(buf.shape) != shape:
                            raise ValueError("saved[%s][%s] must be %s (got %s)" % (name, t, shape, tuple(buf.shape)))
                        getattr(self._sub, 'sv_' + name)[i] = buf.data_ptr()
            self._draw_hop1 = hop1_out is not None
        self.emb = emb if emb is not None else torch.empty((self.n_roots, h2), dtype=torch.float32, device=dev)
        _dev(self.emb, torch.float32, 'emb', 2)
        if tuple(self.emb.shape) != (self.n_roots, h2):
            raise ValueError("emb must be (%d, %d)" % (self.n_roots, h2))
        self.hop1 = torch.empty((self.n_roots, n0), dtype=torch.int32, device=dev) if keep_ids else None
        self.hop2 = torch.empty((self.n_roots, n0 * n1), dtype=torch.int32, device=dev) if keep_ids else None
        if hop_buffers is not None:
            self.hop1, self.hop2 = hop_buffers
            for name, buf, n in (('hop_buffers[0]', self.hop1, self.n_roots * n0), ('hop_buffers[1]', self.hop2, self.n_roots * n0 * n1)):
                if buf is None:                               # (either may be left out)
                    continue
                _dev(buf, torch.int32, name)
                if buf.numel() != n:
                    raise ValueError("%s must have %d entries" % (name, n))
        self._saved = None
        if saved is not None:
            rows = self.n_pairs * (1 + n0)
            shapes = {'x_self': lambda t: (rows, fu if t == 'user' else fi),
                      'x_mean': lambda t: (rows, fi if t == 'user' else fu),
                      'h1': lambda t: (rows, h1), 'x_mean1': lambda t: (self.n_pairs, h1),
                      'z': lambda t: (self.n_pairs, h2)}
            st = _lib.Sage2Saved()
            if saved.get('self_ids') is not None:          # node ids instead of the gathered own rows
                for i, t in enumerate(('user', 'item')):
                    buf = saved['self_ids'][t]
                    _dev(buf, torch.int32, 'saved[self_ids][%s]' % t, 1)
                    if buf.numel() != rows:
                        raise ValueError("saved[self_ids][%s] must have %d entries" % (t, rows))
                    st.self_ids[i] = buf.data_ptr()
                shapes.pop('x_self')
            for name, shape in shapes.items():
                for i, t in enumerate(('user', 'item')):
                    buf = saved[name][t]
                    _dev(buf, torch.float32, 'saved[%s][%s]' % (name, t), 2)
                    if tuple(buf.shape) != shape(t):
                        raise ValueError("saved[%s][%s] must be %s (got %s)" % (name, t, shape(t), tuple(buf.shape)))
                    getattr(st, name)[i] = buf.data_ptr()
            if saved.get('h1_mask') is not None:           # relu masks of the hop-1 heads instead of their h1 rows
                for i, t in enumerate(('user', 'item')):
                    buf = saved['h1_mask'][t]
                    _dev(buf, torch.int32, 'saved[h1_mask][%s]' % t, 1)
                    if buf.numel() != self.n_pairs * n0 * (h1 // 32):
                        raise ValueError("saved[h1_mask][%s] must have %d words" % (t, self.n_pairs * n0 * (h1 // 32)))
                    st.h1_mask[i] = buf.data_ptr()
            rate = float(saved.get('dropout', 0.0) or 0.0)
            if rate:
                if not 0.0 < rate < 1.0:
                    raise ValueError("dropout rate must lie in [0, 1)")
                if saved.get('self_ids') is not None or saved.get('h1_mask') is None:
                    raise ValueError("dropout needs saved['x_self'] (not self_ids) and saved['h1_mask']")
                st.dropout_rate = rate
            self._saved = st
        self._lib = lib
        self._keep = [indptr, indices, feat_user, feat_item, weights, saved]
        self._tape_len = n0 + n0 * n1
        self._fixed_a = (ptr(indptr), ptr(indices), nu + ni, ptr(feat_user), nu, fu, ptr(feat_item), ni, fi)
        self._fixed_b = (ctypes.byref(self._w[0]), ctypes.byref(self._w[1]), ctypes.byref(self._w[2]),
                         ctypes.byref(self._w[3]), h1, h2, int(bool(normalize)), ptr(self.emb), ptr(self.hop1),
                         ptr(self.hop2))
        self._n0, self._n1 = n0, n1

    def _read_weight_addresses(self):
        i = 0
        for layer in (0, 1):
            for t in ('user', 'item'):
                w_self, w_neigh, bias = self._weights[(layer, t)]
                self._w[i].w_self, self._w[i].w_neigh, self._w[i].bias = ptr(w_self), ptr(w_neigh), ptr(bias)
                i += 1
        self._epoch = _storage_epoch

    def run(self, seeds, err_flag, rng_seed=0, walk_id_base=0, walk_ids=None, replay=None):
        if self._epoch != _storage_epoch:
            self._read_weight_addresses()
        if self._saved is not None and self._saved.dropout_rate:
            self._saved.dropout_seed = dropout_seed_of(rng_seed)
        _dev(seeds, torch.int32, 'seeds', 1)
        _dev(err_flag, torch.int32, 'err_flag', 1)
        draw_hop1 = self._sub is not None and self._draw_hop1
        if seeds.numel() != (2 * self.n_pairs if draw_hop1 else self.n_roots):
            raise ValueError("seeds must hold %d roots" % (2 * self.n_pairs if draw_hop1 else self.n_roots))
        if walk_ids is not None:
            _dev(walk_ids, torch.int64, 'walk_ids', 1)
            if walk_ids.numel() != 2 * self.n_pairs:
                raise ValueError("walk_ids must have one entry per seed walk")
        if replay is not None:
            _dev(replay, torch.int32, 'replay', 1)
            words = 2 * self.n_pairs * (self._sub.tape_len if self._sub is not None else self._tape_len)
            if replay.numel() != words:
                raise ValueError("replay tape has the wrong length")
        if self._sub is not None:
            self._sub.seeds = ptr(seeds) if draw_hop1 else None
            check(self._lib.gs_sage2_forward_subtrees(*self._fixed_a, None if draw_hop1 else ptr(seeds), self.n_pairs, self._n0, self._n1,
                                                      int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(walk_id_base) & 0xFFFFFFFFFFFFFFFF,
                                                      ptr(walk_ids), ptr(replay), *self._fixed_b[:6], ptr(self.emb),
                                                      ptr(self.hop1), ptr(self.hop2), ctypes.byref(self._sub),
                                                      ptr(err_flag), stream_handle()),
                  'gs_sage2_forward_subtrees')
            return self.emb
        check(self._lib.gs_sage2_forward(*self._fixed_a, ptr(seeds), self.n_pairs, self._n0, self._n1,
                                         int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(walk_id_base) & 0xFFFFFFFFFFFFFFFF,
                                         ptr(walk_ids), ptr(replay), *self._fixed_b,
                                         ctypes.byref(self._saved) if self._saved is not None else None,
                                         ptr(err_flag), stream_handle()),
              'gs_sage2_forward')
        return self.emb


def sage2_subtrees_supported(fu, fi, h1, h2, n1, n2):
    return bool(load().gs_sage2_subtrees_supported(int(fu), int(fi), int(h1), int(h2), int(n1), int(n2)))


def sage3_seeds_supported(fu, fi, h1, h2, h3, n0):
    return bool(load().gs_sage3_seeds_supported(int(fu), int(fi), int(h1), int(h2), int(h3), int(n0)))


class Sage3SeedsCall:
    """A prepared ``gs_sage3_seeds_forward`` call: layers 0, 1, 2 at the seed nodes of a three-hop
    model on the hop-1 nodes' rows that ``Sage2Call(subtrees=...)`` left.

    ``weights`` = {(layer, node_type): (w_self, w_neigh, bias or None)} for layers 0, 1, 2.
    """

    def __init__(self, feat_user, feat_item, hop1_ids, n0, h0_hop1, h1_hop1, weights, act, normalize, n_pairs, emb,
                 saved=None):
        """``saved`` (training): {'x_mean0', 'h0', 'x_mean1', 'h1', 'x_mean2', 'z'} -> {'user': tensor, 'item': tensor}, one
        row per seed: ``gs_sage3_seeds_forward_saving``."""
        lib = load()
        _dev(feat_user, torch.float32, 'feat_user', 2)
        _dev(feat_item, torch.float32, 'feat_item', 2)
        _dev(hop1_ids, torch.int32, 'hop1_ids')
        _dev(h0_hop1, torch.float32, 'h0_hop1', 2)
        _dev(h1_hop1, torch.float32, 'h1_hop1', 2)
        _dev(emb, torch.float32, 'emb', 2)
        (nu, fu), (ni, fi) = feat_user.shape, feat_item.shape
        n0, n_pairs = int(n0), int(n_pairs)
        self._w, self._weights, self._epoch = [], weights, _storage_epoch
        widths = []
        for layer in (0, 1, 2):
            for t in ('user', 'item'):
                w_self, w_neigh, bias = weights[(layer, t)]
                _dev(w_self, torch.float32, 'w_self', 2)
                _dev(w_neigh, torch.float32, 'w_neigh', 2)
                if bias is not None:
                    _dev(bias, torch.float32, 'bias', 1)
                self._w.append(_lib.SageWeights(ptr(w_self), ptr(w_neigh), ptr(bias)))
            widths.append(2 * weights[(layer, 'user')][0].shape[1])
        h1, h2, h3 = widths
        want = {(0, 'user'): (fu, fi, h1), (0, 'item'): (fi, fu, h1), (1, 'user'): (h1, h1, h2), (1, 'item'): (h1, h1, h2),
                (2, 'user'): (h2, h2, h3), (2, 'item'): (h2, h2, h3)}
        got = {k: (w[0].shape[0], w[1].shape[0], 2 * w[0].shape[1]) for k, w in weights.items()}
        if got != want:
            raise ValueError("weight shapes %s do not fit the model %s" % (got, want))
        if not lib.gs_sage3_seeds_supported(fu, fi, h1, h2, h3, n0):
            raise _lib.GsageError("the seeds kernel does not cover Fu=%d Fi=%d H=[%d,%d,%d] fan-out %d" % (fu, fi, h1, h2, h3, n0))
        rows = 2 * n_pairs * n0
        if hop1_ids.numel() != rows or tuple(h0_hop1.shape) != (rows, h1) or tuple(h1_hop1.shape) != (rows, h2):
            raise ValueError("hop1_ids / h0_hop1 / h1_hop1 must have %d rows (widths %d, %d)" % (rows, h1, h2))
        if tuple(emb.shape) != (2 * n_pairs, h3):
            raise ValueError("emb must be (%d, %d)" % (2 * n_pairs, h3))
        self.n_pairs, self.emb, self._lib = n_pairs, emb, lib
        self._keep = [feat_user, feat_item, hop1_ids, h0_hop1, h1_hop1]
        self._head = (ptr(feat_user), nu, fu, ptr(feat_item), ni, fi)
        self._tail = (n_pairs, ptr(hop1_ids), n0, ptr(h0_hop1), ptr(h1_hop1), *[ctypes.byref(w) for w in self._w],
                      h1, h2, h3, ACT[act], int(bool(normalize)), ptr(emb))
        self._saved = None
        if saved is not None:
            st = _lib.Sage3SeedsSaved()
            for i, t in enumerate(('user', 'item')):
                fo = fi if t == 'user' else fu
                for name, width in (('x_mean0', fo), ('h0', h1), ('x_mean1', h1), ('h1', h2), ('x_mean2', h2), ('z', h3)):
                    buf = saved[name][t]
                    _dev(buf, torch.float32, 'saved[%s][%s]' % (name, t), 2)
                    if tuple(buf.shape) != (n_pairs, width):
                        raise ValueError("saved[%s][%s] must be %s (got %s)" % (name, t, (n_pairs, width), tuple(buf.shape)))
                    getattr(st, name)[i] = buf.data_ptr()
            self._saved = st
            self._keep.append(saved)

    def run(self, seeds):
        if self._epoch != _storage_epoch:
            i = 0
            for layer in (0, 1, 2):
                for t in ('user', 'item'):
                    w_self, w_neigh, bias = self._weights[(layer, t)]
                    self._w[i].w_self, self._w[i].w_neigh, self._w[i].bias = ptr(w_self), ptr(w_neigh), ptr(bias)
                    i += 1
            self._epoch = _storage_epoch
        _dev(seeds, torch.int32, 'seeds', 1)
        if seeds.numel() != 2 * self.n_pairs:
            raise ValueError("seeds must hold %d nodes" % (2 * self.n_pairs))
        if self._saved is not None:
            check(self._lib.gs_sage3_seeds_forward_saving(*self._head, ptr(seeds), *self._tail, ctypes.byref(self._saved),
                                                          stream_handle()), 'gs_sage3_seeds_forward_saving')
            return self.emb
        check(self._lib.gs_sage3_seeds_forward(*self._head, ptr(seeds), *self._tail, stream_handle()), 'gs_sage3_seeds_forward')
        return self.emb


def link_head(emb_user, emb_item, kernel, bias, pred=None):
    """pred = sigmoid(concat(u, i) @ kernel + bias), shape (B,)."""
    lib = load()
    _dev(emb_user, torch.float32, 'emb_user', 2)
    _dev(emb_item, torch.float32, 'emb_item', 2)
    b, e = emb_user.shape
    if tuple(emb_item.shape) != (b, e) or kernel.numel() != 2 * e or bias.numel() != 1:
        raise ValueError("head shapes do not match")
    _dev(kernel, torch.float32, 'kernel')
    _dev(bias, torch.float32, 'bias')
    if pred is None:
        pred = torch.empty(b, dtype=torch.float32, device=emb_user.device)
    check(lib.gs_link_head_fwd(ptr(emb_user), ptr(emb_item), b, e, ptr(kernel), ptr(bias), ptr(pred),
                               stream_handle()), 'gs_link_head_fwd')
    return pred


class LinkHeadStep:
    """Prepared ``gs_link_head_mse_fwd_bwd``: prediction, MSE loss and all head
    gradients in one pass over preallocated buffers."""

    def __init__(self, batch, emb_dim, kernel, bias, g_kernel, g_bias, device):
        lib = load()
        for name, t, n in (('kernel', kernel, 2 * emb_dim), ('bias', bias, 1), ('g_kernel', g_kernel, 2 * emb_dim),
                           ('g_bias', g_bias, 1)):
            _dev(t, torch.float32, name)
            if t.numel() != n:
                raise ValueError("%s must have %d elements" % (name, n))
        self.b, self.e = int(batch), int(emb_dim)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        self.pred, self.loss = new(self.b), new(1)
        self.g_user, self.g_item = new(self.b, self.e), new(self.b, self.e)
        self.ws = new(max(lib.gs_link_head_workspace(self.b, self.e) // 4, 1))
        self._lib = lib
        self._keep = (kernel, bias, g_kernel, g_bias)
        self.epoch = _storage_epoch

    def run(self, emb_user, emb_item, targets):
        if self.epoch != _storage_epoch:
            raise _lib.GsageError("parameters were moved to other storage after this call was prepared; prepare it again")
        kernel, bias, g_kernel, g_bias = self._keep
        for name, t, shape in (('emb_user', emb_user, (self.b, self.e)), ('emb_item', emb_item, (self.b, self.e))):
            _dev(t, torch.float32, name, 2)
            if tuple(t.shape) != shape:
                raise ValueError("%s must be %s" % (name, shape))
        _dev(targets, torch.float32, 'targets')
        if targets.numel() != self.b:
            raise ValueError("targets must have %d entries" % self.b)
        check(self._lib.gs_link_head_mse_fwd_bwd(ptr(emb_user), ptr(emb_item), self.b, self.e, ptr(kernel), ptr(bias),
                                                 ptr(targets), ptr(self.pred), ptr(self.loss), ptr(self.g_user),
                                                 ptr(self.g_item), ptr(g_kernel), ptr(g_bias), ptr(self.ws),
                                                 self.ws.numel() * 4, stream_handle()), 'gs_link_head_mse_fwd_bwd')
        return self.loss, self.pred


def adam_step(params_flat, grads_flat, m, v, lr_t, beta1, beta2, eps, lr_t_device=None, grad_scale=1.0):
    """lr_t_device: optional 1-element GPU tensor holding the rate (read by the kernel);
    grad_scale multiplies every gradient as it is read (1 / world after a summing all-reduce)."""
    lib = load()
    if lr_t_device is not None:
        _dev(lr_t_device, torch.float32, 'lr_t_device')
    for name, t in (('params', params_flat), ('grads', grads_flat), ('m', m), ('v', v)):
        _dev(t, torch.float32, name, 1)
        if t.numel() != params_flat.numel():
            raise ValueError("flat optimizer buffers differ in length")
    check(lib.gs_adam_step(ptr(params_flat), ptr(grads_flat), ptr(m), ptr(v), params_flat.numel(), float(lr_t),
                           ptr(lr_t_device), float(beta1), float(beta2), float(eps), float(grad_scale), stream_handle()),
          'gs_adam_step')


def stage_batch(seeds, seeds_dst, walk_ids, walk_ids_dst, targets, targets_dst, lr_t=0.0, lr_dst=None):
    """``gs_stage_batch``: a step's inputs into the buffers its captured graph reads
    (and the step's Adam rate into ``lr_dst``), one launch.  Sources must already
    have the destination's dtype, length and device."""
    lib = load()
    for (src, dst, dtype, name) in ((seeds, seeds_dst, torch.int32, 'seeds'), (walk_ids, walk_ids_dst, torch.int64, 'walk_ids'),
                                    (targets, targets_dst, torch.float32, 'targets')):
        _dev(src, dtype, name)
        _dev(dst, dtype, name + '_dst')
        if src.numel() != dst.numel():
            raise ValueError("%s has %d elements, its buffer %d" % (name, src.numel(), dst.numel()))
    if lr_dst is not None:
        _dev(lr_dst, torch.float32, 'lr_dst')
    check(lib.gs_stage_batch(ptr(seeds), ptr(seeds_dst), seeds.numel(), ptr(walk_ids), ptr(walk_ids_dst), walk_ids.numel(),
                             ptr(targets), ptr(targets_dst), targets.numel(), float(lr_t), ptr(lr_dst), stream_handle()),
          'gs_stage_batch')
