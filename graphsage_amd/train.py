"""Seed-edge data-parallel training of the link model.

The reference trains through Keras: ``concat(user_emb, item_emb) -> Dense(1,
sigmoid)`` (``link_utils.py:167-168,244-247``), mean-squared-error loss and
``Adam(lr=1e-2)`` (``train_movielens.py:183-189``), one process.  Here the same
step runs one process per GPU: every rank holds the whole graph and feature
tables, takes its own shard of the seed edges (``flow(rank=, world_size=)``),
runs sampling + forward + backward through the HIP kernels, and the ranks
average ALL gradients with a single RCCL all-reduce over one flat fp32 bucket
(a few thousand to a few hundred thousand floats: latency-bound, so one
collective per step is the right granularity on xGMI).

The link head and the optimizer update are a handful of tiny dense ops and use
torch tensor arithmetic on the GPU.
"""
import math

import torch

__all__ = ['LinkHead', 'FlatGradBucket', 'Adam', 'Trainer']


class LinkHead:
    """concat -> Dense(1) -> sigmoid, Keras initialisers (glorot-uniform kernel,
    zero bias); output shape (B, 1) as after ``Reshape((1,))``."""

    def __init__(self, emb_dim, device, seed=None):
        gen = torch.Generator(device='cpu')
        if seed is None:
            gen.seed()
        else:
            gen.manual_seed(int(seed))
        limit = math.sqrt(6.0 / (2 * emb_dim + 1))
        k = (torch.rand(2 * emb_dim, 1, generator=gen, dtype=torch.float64) * 2 - 1) * limit
        self.kernel = k.to(torch.float32).to(device).requires_grad_(True)
        self.bias = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)

    def parameters(self):
        return [self.kernel, self.bias]

    def __call__(self, x_out):
        user_emb, item_emb = x_out
        z = torch.cat([user_emb, item_emb], dim=-1) @ self.kernel + self.bias
        return torch.sigmoid(z).reshape(-1, 1)


class FlatGradBucket:
    """All gradients in one contiguous fp32 buffer.

    ``p.grad`` of every parameter is a view into the buffer, so backward writes
    straight into it and the all-reduce needs no packing or unpacking.
    """

    def __init__(self, params):
        self.params = list(params)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        """Sum over ranks, divide by the world size (no-op for a single process)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return self.flat
        world = dist.get_world_size(group)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(world)
        return self.flat


class Adam:
    """Keras-default Adam on one flat parameter/gradient pair: beta1 0.9, beta2
    0.999, epsilon 1e-7, no amsgrad; ``lr_t = lr * sqrt(1-b2^t) / (1-b1^t)``,
    ``p -= lr_t * m / (sqrt(v) + eps)`` (the TF2 formulation)."""

    def __init__(self, params, grads_flat, lr=1e-2, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.params = list(params)
        self.g = grads_flat
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.m = torch.zeros_like(grads_flat)
        self.v = torch.zeros_like(grads_flat)
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        self.m.mul_(self.b1).add_(self.g, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(self.g, self.g, value=1 - self.b2)
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        upd = self.m / (self.v.sqrt() + self.eps)
        off = 0
        for p in self.params:
            p.add_(upd[off:off + p.numel()].view_as(p), alpha=-lr_t)
            off += p.numel()


class Trainer:
    """One optimisation step = batch -> embeddings -> head -> MSE -> backward ->
    gradient all-reduce -> Adam."""

    def __init__(self, model, head=None, lr=1e-2, group=None, engine='auto', graph=None, batch_size=None):
        """engine: 'autograd' runs the model through torch.autograd Functions (any
        shape); 'plan' uses the static launch plan of ``SageTrainEngine`` (needs
        ``graph`` and ``batch_size``); 'auto' picks the plan when it applies."""
        self.model = model.build()
        dev = self.model.parameters()[0].device
        emb_dim = self.model.dims[-1]['user']
        self.head = head or LinkHead(emb_dim, dev)
        self.params = self.model.parameters() + self.head.parameters()
        self.bucket = FlatGradBucket(self.params)
        self.opt = Adam(self.params, self.bucket.flat, lr=lr)
        self.group = group
        self.engine = None
        if engine in ('auto', 'plan') and graph is not None and batch_size is not None:
            from .train_engine import SageTrainEngine
            try:
                self.engine = SageTrainEngine(graph, self.model, batch_size)
            except Exception:
                if engine == 'plan':
                    raise
        elif engine == 'plan':
            raise ValueError("engine='plan' needs graph and batch_size")
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Rank 0's initial weights everywhere (ranks seed their own generators)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            with torch.no_grad():
                flat = torch.cat([p.reshape(-1) for p in self.params])
                dist.broadcast(flat, src=0, group=self.group)
                off = 0
                for p in self.params:
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()

    def loss(self, batch, targets):
        pred = self.head(self.model(batch))
        y = torch.as_tensor(targets, dtype=torch.float32, device=pred.device).reshape(pred.shape)
        return ((pred - y) ** 2).mean(), pred

    def step_device(self, seeds, targets, walk_ids=None, rng_seed=0):
        """Training step on device-resident inputs through the static plan:
        ``seeds`` int32[2B] roots (sources then destinations), ``targets``
        float32[B] on the GPU."""
        eng = self.engine
        self.bucket.zero()
        emb_u, emb_i = eng.forward_train(seeds, walk_ids, rng_seed)
        emb_u = emb_u.detach().requires_grad_(True)
        emb_i = emb_i.detach().requires_grad_(True)
        pred = self.head([emb_u, emb_i])
        loss = ((pred - targets.reshape(pred.shape)) ** 2).mean()
        loss.backward()                                   # head gradients + embedding gradients
        eng.backward(emb_u.grad, emb_i.grad)
        self.bucket.all_reduce_mean(self.group)
        self.opt.step()
        return loss.detach(), pred.detach()

    def step(self, batch, targets):
        self.bucket.zero()
        loss, pred = self.loss(batch, targets)
        loss.backward()
        self.bucket.all_reduce_mean(self.group)
        self.opt.step()
        return loss.detach(), pred.detach()
