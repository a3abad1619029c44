"""MI355X-native bipartite GraphSAGE hot path: sampling, gather-mean aggregation
and per-type projection as HIP kernels behind the reference's Python classes."""
from .bipartite_graph import Graph, IsolatedNodeError
from .sampler import BreadthFirstWalker
from .generator import (BatchFeatures, BatchedLinkGenerator, Generator, GraphSAGEGenerator,
                        LinkSequence)
from .utils import Aggregator, is_real_iterable, random_state
from .models import GraphSAGE

__all__ = ['Graph', 'IsolatedNodeError', 'BreadthFirstWalker', 'BatchFeatures', 'BatchedLinkGenerator',
           'Generator', 'GraphSAGEGenerator', 'LinkSequence', 'Aggregator', 'is_real_iterable',
           'random_state', 'GraphSAGE']
