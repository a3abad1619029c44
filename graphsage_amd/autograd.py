"""torch.autograd glue: records which HIP kernel undoes which, nothing more.

Each Function calls one forward entry point of ``libgsage.so`` and, in
``backward``, the matching backward entry point.  Feature tables are constants
of the reference model (they are inputs, not variables, ``models.py:241``), so
the indexed gather-mean only propagates a gradient into the table when the
table itself requires one (trainable embeddings); then it is the float-atomic
scatter-add kernel.
"""
import torch

from . import kernels


class GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, ids, id_offset):
        ctx.save_for_backward(ids)
        ctx.meta = (feat.shape, int(id_offset))
        return kernels.gather_rows(feat, ids, id_offset)

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        (ids,) = ctx.saved_tensors
        shape, off = ctx.meta
        g_feat = torch.zeros(shape, dtype=torch.float32, device=g.device)
        kernels.gather_mean_bwd(g.contiguous(), ids, 1, g_feat, off)
        return g_feat, None, None


class GatherMean(torch.autograd.Function):
    """mean over the n_child sampled children of each parent, rows taken from a
    feature table by node id."""

    @staticmethod
    def forward(ctx, feat, child_ids, n_parents, n_child, id_offset):
        ctx.save_for_backward(child_ids)
        ctx.meta = (feat.shape, int(n_child), int(id_offset))
        return kernels.gather_mean(feat, child_ids, n_parents, n_child, id_offset)

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        (child_ids,) = ctx.saved_tensors
        shape, n_child, off = ctx.meta
        g_feat = torch.zeros(shape, dtype=torch.float32, device=g.device)
        kernels.gather_mean_bwd(g.contiguous(), child_ids, n_child, g_feat, off)
        return g_feat, None, None, None, None


class GroupMean(torch.autograd.Function):
    """mean over axis 1 of a dense (n_parents, n_child, F) tensor."""

    @staticmethod
    def forward(ctx, x, n_parents, n_child):
        ctx.meta = (x.shape, int(n_parents), int(n_child))
        return kernels.gather_mean(x.reshape(n_parents * n_child, -1), None, n_parents, n_child)

    @staticmethod
    def backward(ctx, g):
        shape, n_parents, n_child = ctx.meta
        g_x = torch.empty((n_parents * n_child, g.shape[1]), dtype=torch.float32, device=g.device)
        kernels.gather_mean_bwd(g.contiguous(), None, n_child, g_x)
        return g_x.reshape(shape), None, None


class GatherMax(torch.autograd.Function):
    """max over the sampled children of each parent, rows taken from a feature
    table by node id; the table is an input of the model and gets no gradient."""

    @staticmethod
    def forward(ctx, feat, child_ids, n_parents, n_child, id_offset):
        if feat.requires_grad:
            raise NotImplementedError("the max aggregator has no gradient into a feature table")
        return kernels.gather_max(feat, child_ids, n_parents, n_child, id_offset)

    @staticmethod
    def backward(ctx, g):
        return None, None, None, None, None


class GroupMax(torch.autograd.Function):
    """max over axis 1 of a dense (n_parents, n_child, F) tensor; ties share the
    gradient evenly."""

    @staticmethod
    def forward(ctx, x, n_parents, n_child):
        x2 = x.reshape(n_parents * n_child, -1)
        out = kernels.gather_max(x2, None, n_parents, n_child)
        ctx.save_for_backward(x2, out)
        ctx.meta = (x.shape, int(n_child))
        return out

    @staticmethod
    def backward(ctx, g):
        x2, out = ctx.saved_tensors
        shape, n_child = ctx.meta
        return kernels.group_max_bwd(g.contiguous(), x2, out, n_child).reshape(shape), None, None


class SageProject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_self, x_mean, w_self, w_neigh, bias, act):
        out = kernels.sage_project(x_self, x_mean, w_self, w_neigh, bias, act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x_self, x_mean, w_self, w_neigh, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x_self, x_mean, w_self, w_neigh, out = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_ws, g_wn, g_b, g_xs, g_xm = kernels.sage_project_bwd(
            x_self, x_mean, w_self, w_neigh, ctx.act, out, g.contiguous(),
            need_x_self=need[0], need_x_mean=need[1], need_bias=ctx.has_bias)
        return g_xs, g_xm, g_ws, g_wn, g_b, None


class L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return kernels.l2_normalize(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return kernels.l2_normalize_bwd(x, g.contiguous())
