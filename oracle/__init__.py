"""CPU restatement of the reference's bipartite GraphSAGE hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It exists so that the
HIP kernels behind ``graphsage_amd`` have something independent to be checked
against.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; nothing under ``graphsage_amd/`` does, and
the product path raises when its HIP library is missing instead of coming here.

What is restated, and from where (all paths are files of the reference
repository):

* ``graph.py``      - ``bipartite_graph.py:9-93`` (adjacency dicts in edge-column
                      order, item feature row = id - num_users)
* ``philox.py``     - Philox4x32-10 (Salmon et al., SC'11; Random123 1.14
                      constants); the reference has no counter RNG, this is the
                      build's own generator, restated on the CPU so that the
                      device stream can be checked word for word
* ``walker.py``     - ``sampler.py:47-96`` (breadth-first walk, uniform draws
                      with replacement, ValueError on an empty neighbor list)
* ``generator.py``  - ``generator.py:17-110`` (LinkSequence batching/shuffle),
                      ``generator.py:239-244,281-337`` (slot regroup/interlace),
                      ``generator.py:248-279`` + ``bipartite_graph.py:70-76``
                      (feature gather and reshape)
* ``model.py``      - ``utils.py:64-120`` (Aggregator), ``models.py:48-190``
                      (layer wiring, l2 normalisation), ``link_utils.py:167-168,
                      236-247`` (concat + Dense(1, sigmoid) head)

Pinning status: the integer/index part (graph, walker, generator) is pinned by
fixtures under ``tests/golden/`` that were produced by importing the reference's
own ``bipartite_graph.py``, ``sampler.py`` and ``generator.py`` (script:
``tests/golden/make_golden.py``).  The floating-point part (``model.py``) is
"parity unpinned": it depends on TensorFlow, which is not installed here and has
no fixtures in the reference, so it is a restatement of the cited lines with
float64 accumulation as arbiter and nothing more.
"""
