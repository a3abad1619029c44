"""Link prediction head factory, the slice of the reference's ``link_utils.py``
its training script uses: ``link_classification(edge_embedding_method="concat")``
(``train_movielens.py:179``) = concatenate the two node embeddings, ``Dense(1,
sigmoid)``, reshape to ``(B, 1)`` (``link_utils.py:167-168,244-247,312-317``).

The other edge-embedding operators of that file (ip, mul, l1, l2, avg), softmax
outputs and the clipped regression head are not on the accelerated path and
raise ``NotImplementedError``.
"""
from .train import LinkHead

__all__ = ['link_inference', 'link_classification']


def link_inference(output_dim=1, output_act="linear", edge_embedding_method="ip", clip_limits=None,
                   name="link_inference"):
    if edge_embedding_method != "concat" or output_dim != 1 or output_act != "sigmoid" or clip_limits:
        raise NotImplementedError(
            "%s: only edge_embedding_method='concat' with Dense(1, sigmoid) is built "
            "(got method=%r, output_dim=%r, output_act=%r, clip_limits=%r)"
            % (name, edge_embedding_method, output_dim, output_act, clip_limits))

    def edge_function(x_out, seed=None):
        """``x_out`` is what ``GraphSAGE.in_out_tensors()`` returns second (the
        built model); the head takes its embedding width and device from it."""
        model = x_out.build()
        return LinkHead(model.dims[-1]['user'], model.parameters()[0].device, seed=seed)

    print("{}: using '{}' method to combine node embeddings into edge embeddings".format(name, edge_embedding_method))
    return edge_function


def link_classification(output_dim=1, output_act="sigmoid", edge_embedding_method="ip"):
    return link_inference(output_dim=output_dim, output_act=output_act,
                          edge_embedding_method=edge_embedding_method, name="link_classification")
