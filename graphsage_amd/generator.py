"""Link batch generator: ``flow()`` -> ``LinkSequence`` -> sampled batches.

Same class names, constructor arguments and error behaviour as the reference
(``generator.py:17-345``).  What changed is where a batch is made:
``GraphSAGEGenerator.sample_features`` launches the sampler kernel once for the
2B roots of a batch (all source heads, then all destination heads -- the order
of ``generator.py:301-309``) and gets, per hop, one dense id array whose first B
rows belong to the source heads and last B rows to the destination heads.
Those halves ARE the type-homogeneous slots the reference assembles with list
concatenation (``:314-333``), so no regroup pass exists.

A batch is returned as ``BatchFeatures``: it indexes like the reference's list
of ``(B, n, F)`` feature arrays (slot tensors are gathered on first access) and
also carries the slot ids, which lets the model fuse gather and neighbor mean
without ever writing the deepest slots to memory.

Seed-edge data parallelism hooks in here: ``flow(..., rank=r, world_size=w)``
gives rank ``r`` every ``w``-th link of the (identically shuffled) epoch order.
"""
import abc
import collections.abc

import numpy as np

from . import kernels
from .sampler import BreadthFirstWalker
from .utils import is_real_iterable, random_state

__all__ = ['LinkSequence', 'Generator', 'BatchedLinkGenerator', 'GraphSAGEGenerator', 'BatchFeatures']

_TYPES = ('user', 'item')
_MASK64 = 0xFFFFFFFFFFFFFFFF


def _mix64(a, b):
    """splitmix64 finaliser over (a, b): derives the per-epoch sampling key."""
    z = (int(a) + 0x9E3779B97F4A7C15 * (int(b) + 1)) & _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


class BatchFeatures(collections.abc.Sequence):
    """One sampled batch: slot ``2*h + side`` holds hop ``h`` of the walks rooted
    at the source (side 0) / destination (side 1) heads.

    ``batch[i]`` is the ``(B, n_i, F_type)`` float32 GPU tensor of slot ``i``
    (what the reference returns as numpy arrays); ``batch.ids[i]`` the int32 node
    ids of the slot, shape ``(B, n_i)``.
    """

    def __init__(self, graph, head_size, slot_ids, slot_types):
        self.graph = graph
        self.head_size = head_size
        self.ids = slot_ids
        self.types = slot_types
        self._feats = [None] * len(slot_ids)

    def __len__(self):
        return len(self.ids)

    def table(self, node_type):
        """(feature table, id offset) of a node type."""
        dev = self.graph.device_tensors()
        return dev[node_type], (0 if node_type == 'user' else self.graph.correction_constant)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if self._feats[i] is None:
            feat, off = self.table(self.types[i])
            rows = kernels.gather_rows(feat, self.ids[i].reshape(-1), off)
            self._feats[i] = rows.reshape(self.head_size, -1, feat.shape[1])
        return self._feats[i]

    def numpy(self):
        return [self[i].cpu().numpy() for i in range(len(self))]


class LinkSequence:
    """Batches of link ids for ``fit``-style loops (``generator.py:17-110``).

    ``rank`` / ``world_size`` shard the epoch order for data-parallel training.
    """

    def __init__(self, sample_function, batch_size, ids, targets=None, shuffle=True, seed=None,
                 rank=0, world_size=1):
        if not is_real_iterable(ids):
            raise TypeError("IDs must be an iterable or numpy array of graph node IDs")
        if targets is not None:
            if not is_real_iterable(targets):
                raise TypeError("Targets must be None or an iterable or numpy array ")
            if len(ids) != len(targets):
                raise ValueError("The length of the targets must be the same as the length of the ids")
            self.targets = np.asanyarray(targets)
        else:
            self.targets = None
        if isinstance(sample_function, collections.abc.Callable):
            self._sample_features = sample_function
        else:
            raise TypeError("({}) The sampling function expects a callable function.".format(type(self).__name__))
        if not 0 <= rank < world_size:
            raise ValueError("rank %d outside world of size %d" % (rank, world_size))
        if world_size > 1 and shuffle and seed is None:
            raise ValueError("data-parallel shuffling needs a seed so that all ranks agree on the epoch order")

        self.batch_size = batch_size
        self.ids = list(ids)
        self.shuffle = shuffle
        self.rank, self.world_size = rank, world_size
        self._rs, _ = random_state(seed)
        self._indexed = getattr(sample_function, '_accepts_link_index', False)
        self.salt = 0                  # set by the generator: distinguishes the draws of its sequences
        self.epoch = -1
        self.on_epoch_end()

    def __len__(self):
        return int(np.ceil(self.data_size / self.batch_size))

    def __getitem__(self, batch_num):
        start_idx = self.batch_size * batch_num
        end_idx = start_idx + self.batch_size
        if start_idx >= self.data_size:
            raise IndexError("Mapper: batch_num larger than length of data")
        batch_indices = self.indices[start_idx:end_idx]
        head_ids = [self.ids[ii] for ii in batch_indices]
        batch_targets = self.targets[batch_indices]
        if self._indexed:
            batch_feats = self._sample_features(head_ids, batch_num, link_index=batch_indices, epoch=self.epoch,
                                                salt=self.salt)
        else:
            batch_feats = self._sample_features(head_ids, batch_num)
        return batch_feats, batch_targets

    def roots(self, batch_num):
        """The batch as the kernels take it, without sampling: (roots int32[2b] on
        the GPU -- sources then destinations --, walk ids int64[2b], sampling key,
        targets).  Only for sequences made by ``GraphSAGEGenerator.flow``."""
        if not self._indexed:
            raise TypeError("this sequence's sampling function does not key draws by link")
        start_idx = self.batch_size * batch_num
        if start_idx >= self.data_size:
            raise IndexError("Mapper: batch_num larger than length of data")
        batch_indices = self.indices[start_idx:start_idx + self.batch_size]
        head_ids = [self.ids[ii] for ii in batch_indices]
        seeds, walk_ids, key = self._sample_features.roots(head_ids, batch_indices, self.epoch, self.salt)
        return seeds, walk_ids, key, (None if self.targets is None else self.targets[batch_indices])

    def epoch_on_device(self):
        """Every FULL batch of the current epoch staged on the GPU with one copy per
        array: dict(roots int32[n, 2b] -- sources then destinations per row --,
        walk_ids int64[n, 2b], targets float32[n, b] or None, key, n_batches).
        A loop that feeds the kernels from these rows pays no per-batch host work."""
        if not self._indexed:
            raise TypeError("this sequence's sampling function does not key draws by link")
        if getattr(self, '_staged_epoch', None) == self.epoch:
            return self._staged
        n_full = self.data_size // self.batch_size
        idx = np.asarray(self.indices[:n_full * self.batch_size], dtype=np.int64)
        self._staged = self._sample_features.stage(self.ids, idx, n_full, self.batch_size, self.targets, self.epoch,
                                                   self.salt)
        self._staged_epoch = self.epoch
        return self._staged

    def __iter__(self):
        for b in range(len(self)):
            yield self[b]

    def on_epoch_end(self):
        order = list(range(len(self.ids)))
        if self.shuffle:
            self._rs.shuffle(order)
        self.indices = order[self.rank::self.world_size] if self.world_size > 1 else order
        self.data_size = len(self.indices)
        self.epoch += 1


class Generator(abc.ABC):
    @abc.abstractmethod
    def num_batch_dims(self):
        pass

    @abc.abstractmethod
    def flow(self, *args, **kwargs):
        pass


class BatchedLinkGenerator(Generator):
    def __init__(self, Graph, batch_size):
        self.graph = Graph
        self.batch_size = batch_size
        self.multiplicity = 2
        self.head_node_types = None
        self.sampler = None

    @abc.abstractmethod
    def sample_features(self, head_links, batch_num):
        pass

    def num_batch_dims(self):
        return 1

    def flow(self, link_ids, targets=None, shuffle=False, seed=None, rank=0, world_size=1):
        """Sequence over ``link_ids`` (pairs ``(source, destination)``).

        As in the reference (``generator.py:171-204``) only the first link is
        validated -- arity (``KeyError``), membership (``KeyError``), head types
        (``ValueError``) -- and an empty ``link_ids`` yields ``None``.  Later
        links that name unknown nodes surface as ``KeyError`` when their batch
        is sampled.
        """
        for link in link_ids:
            if len(link) != 2:
                raise KeyError("Expected link IDs to be a tuple of length 2")
            src, dst = link
            try:
                node_type_src = self.graph.get_node_type(src)
            except KeyError:
                raise KeyError(f"Node ID {src} supplied to generator not found in graph")
            try:
                node_type_dst = self.graph.get_node_type(dst)
            except KeyError:
                raise KeyError(f"Node ID {dst} supplied to generator not found in graph")
            if self.head_node_types is not None and (
                    node_type_src != self.head_node_types[0] or node_type_dst != self.head_node_types[1]):
                raise ValueError(
                    f"Node pair ({src}, {dst}) not of expected type "
                    f"({self.head_node_types[0]}, {self.head_node_types[1]})")
            return LinkSequence(self.sample_features, self.batch_size, link_ids, targets=targets,
                                shuffle=shuffle, seed=seed, rank=rank, world_size=world_size)
        return None


class _IndexedSampler:
    """Bound ``sample_features`` that also accepts the links' positions in the
    id list and the epoch, so draws are keyed by link, not by batch layout."""
    _accepts_link_index = True

    def __init__(self, gen):
        self._gen = gen

    def __call__(self, head_links, batch_num, link_index=None, epoch=0, salt=0):
        return self._gen._sample(head_links, link_index, epoch, salt)

    def roots(self, head_links, link_index, epoch, salt=0):
        roots, walk_ids, _ = self._gen._roots(head_links, link_index)
        return roots, walk_ids, self._gen.sampling_key(epoch, salt)

    def stage(self, ids, idx, n_full, b, targets, epoch, salt=0):
        import torch
        from ._lib import require_gpu
        dev = require_gpu()
        gen = self._gen
        if getattr(self, '_links_src', None) is not ids:            # the id list as an array, made once
            self._links_np = np.asarray([(l[0], l[1]) for l in ids], dtype=np.int64).reshape(-1, 2)
            self._links_src = ids
        links = self._links_np[idx]
        n = gen.graph.num_users + gen.graph.num_items
        if links.size and (links.min() < 0 or links.max() >= n):
            raise KeyError(int(links[(links < 0) | (links >= n)][0]))
        present = getattr(gen.graph, '_present', None)
        if links.size and present is not None and not present[links].all():
            raise KeyError(int(links[~present[links]][0]))
        per = links.reshape(n_full, b, 2)
        roots = np.concatenate([per[:, :, 0], per[:, :, 1]], axis=1).astype(np.int32)
        li = idx.reshape(n_full, b)
        walk = np.concatenate([2 * li, 2 * li + 1], axis=1)
        out = {'roots': torch.from_numpy(np.ascontiguousarray(roots)).to(dev),
               'walk_ids': torch.from_numpy(np.ascontiguousarray(walk)).to(dev),
               'targets': None, 'key': gen.sampling_key(epoch, salt), 'n_batches': n_full}
        if targets is not None:
            out['targets'] = torch.from_numpy(np.asarray(targets, dtype=np.float32)[idx].reshape(n_full, b)).to(dev)
        return out


class GraphSAGEGenerator(BatchedLinkGenerator):
    def __init__(self, Graph, batch_size, num_samples, head_node_types=None, seed=None):
        """``seed`` keys the Philox sampler (the reference accepts and ignores
        it, ``generator.py:215``)."""
        super().__init__(Graph, batch_size)
        self.num_samples = num_samples
        self.head_node_types = head_node_types
        self.seed = seed
        self.sampler = BreadthFirstWalker(Graph=Graph, num_of_walks=num_samples, seed=seed)
        n_hops = len(num_samples)
        # slot s = 2*h + side: node type alternates with the hop
        self._slot_types = [_TYPES[(s + h) % 2] for h in range(n_hops + 1) for s in range(2)]
        self._sampling_schema = self._schema(num_samples)
        self._indexed = _IndexedSampler(self)
        self._n_flows = 0

    @staticmethod
    def _schema(num_samples):
        """For each head side, per slot: (type, walk-list positions that feed it);
        the two-hop case spells out ``generator.py:239-244``."""
        sizes = [1] + kernels.hop_sizes(num_samples)
        first, groups = [0, 1], [1, 1]
        for h in range(2, len(num_samples) + 1):
            first.append(first[-1] + groups[-1])
            groups.append(sizes[h - 1])
        schema = []
        for side in range(2):
            schema.append([(_TYPES[(s + h) % 2],
                            list(range(first[h], first[h] + groups[h])) if s == side else [])
                           for h in range(len(num_samples) + 1) for s in range(2)])
        return schema

    @property
    def type_adjacency_list(self):
        """(type, [neighbor slot]) per slot; two hops give ``generator.py:341-344``."""
        n_hops = len(self.num_samples)
        return [(self._slot_types[2 * h + s], [2 * (h + 1) + s] if h < n_hops else [])
                for h in range(n_hops + 1) for s in range(2)]

    def flow(self, link_ids, targets=None, shuffle=False, seed=None, rank=0, world_size=1):
        seq = super().flow(link_ids, targets=targets, shuffle=shuffle, seed=seed, rank=rank,
                           world_size=world_size)
        if seq is not None:
            seq._sample_features = self._indexed
            seq._indexed = True
            # The reference draws all its sequences' samples from one global stream, so link i of the
            # training sequence and link i of the validation sequence get unrelated neighborhoods.
            # Here draws are addressed by link position; the sequence's ordinal goes into the key so
            # that the same holds.  Ranks of a data-parallel job call flow() in the same order.
            seq.salt = self._n_flows
            self._n_flows += 1
        return seq

    def sampling_key(self, epoch, salt=0):
        """Philox key of one epoch of one sequence."""
        return _mix64(_mix64(self._base_seed(), salt), epoch)

    def sample_features(self, head_links, batch_num):
        """Reference signature (``generator.py:281``): walk ids continue from the
        sampler's running counter, like a global random stream advancing."""
        return self._sample(head_links, None, 0)

    def _roots(self, head_links, link_index):
        """(roots on the GPU, walk ids or None, batch size) after the membership
        checks the reference makes through its dict lookups."""
        import torch
        from ._lib import require_gpu
        dev = require_gpu()
        links = np.asarray([(l[0], l[1]) for l in head_links], dtype=np.int64).reshape(-1, 2)
        b = links.shape[0]
        n = self.graph.num_users + self.graph.num_items
        if b and (links.min() < 0 or links.max() >= n):
            bad = links[(links < 0) | (links >= n)][0]
            raise KeyError(int(bad))
        present = getattr(self.graph, '_present', None)
        if b and present is not None and not present[links].all():
            raise KeyError(int(links[~present[links]][0]))
        roots = torch.from_numpy(np.concatenate([links[:, 0], links[:, 1]]).astype(np.int32)).to(dev)
        walk_ids = None
        if link_index is not None:
            li = np.asarray(link_index, dtype=np.int64)
            walk_ids = torch.from_numpy(np.concatenate([2 * li, 2 * li + 1])).to(dev)
        return roots, walk_ids, b

    def _sample(self, head_links, link_index, epoch, salt=0):
        roots, walk_ids, b = self._roots(head_links, link_index)
        # keyed sequences bring their own key; the walker's seed (running-counter calls) is not touched
        key = self.sampling_key(epoch, salt) if walk_ids is not None else None
        hops = self.sampler.sample_device(roots, walk_ids=walk_ids, rng_seed=key)
        slot_ids = [roots[:b].reshape(b, 1), roots[b:].reshape(b, 1)]
        for hop in hops:
            slot_ids.append(hop[:b])
            slot_ids.append(hop[b:])
        return BatchFeatures(self.graph, b, slot_ids, self._slot_types)

    def _base_seed(self):
        return 0x9E3779B97F4A7C15 if self.seed is None else int(self.seed)
