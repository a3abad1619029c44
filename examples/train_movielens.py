#!/usr/bin/env python3
"""MovieLens link prediction end to end on one MI355X (or one process per GPU
under torchrun), the run of the reference's ``train_movielens.py``: preprocess ->
Graph -> GraphSAGEGenerator.flow -> GraphSAGE([32, 32]) -> concat/Dense(1,
sigmoid) head -> Adam(1e-2) on MSE -> loss / RMSE / accuracy / AUC.

    python examples/train_movielens.py --data-dir /path/with/users.dat+movies.dat \\
        [--ratings /path/ratings.dat | --synthetic-ratings 200000]

ratings.dat is not shipped with the reference; without it a MovieLens-shaped
synthetic rating table over the real users and movies is drawn.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--data-dir', required=True)
    ap.add_argument('--ratings', default=None)
    ap.add_argument('--synthetic-ratings', type=int, default=200_000)
    ap.add_argument('--batch-size', type=int, default=100)          # train_movielens.py:16
    ap.add_argument('--num-samples', type=int, nargs='+', default=[8, 4])   # :157
    ap.add_argument('--layer-sizes', type=int, nargs='+', default=[32, 32])  # :175
    ap.add_argument('--epochs', type=int, default=1)                 # :192
    ap.add_argument('--lr', type=float, default=1e-2)                # :185
    ap.add_argument('--pad-features', type=int, default=0,
                    help='pad the feature tables to a multiple of this at load time (0: raw widths; the engines pad internally)')
    ap.add_argument('--reference-feature-order', action='store_true',
                    help="keep feature rows in file order as the reference's script does")
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from graphsage_amd import Graph, GraphSAGE, GraphSAGEGenerator, movielens
    from graphsage_amd.link_utils import link_classification
    from graphsage_amd.train import Trainer

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    if world > 1:
        dist.init_process_group('nccl')

    users, movies = os.path.join(args.data_dir, 'users.dat'), os.path.join(args.data_dir, 'movies.dat')
    ratings = args.ratings or os.path.join(args.data_dir, 'ratings.dat')
    if not os.path.exists(ratings):
        print("no ratings.dat: drawing %d synthetic ratings" % args.synthetic_ratings)
        ratings = movielens.synthetic_ratings(users, movies, args.synthetic_ratings, seed=args.seed)
    data = movielens.load_movielens(users, movies, ratings, align_features=not args.reference_feature_order,
                                    pad_to=args.pad_features or None)
    graph = Graph(node_features=data['node_features'], nodes=data['nodes'], edges=data['edges'])
    print(graph, "- features", graph.node_feature_sizes, "- links", len(data['links']))
    print("Positive Value Ratio: {:.4f}".format(float(np.mean(data['labels'] == 1))))

    train_idx, test_idx = movielens.train_test_split(len(data['links']), 0.7, 0.3, seed=args.seed)
    links = [tuple(l) for l in data['links'].tolist()]
    generator = GraphSAGEGenerator(graph, args.batch_size, args.num_samples, head_node_types=["user", "item"],
                                   seed=args.seed)
    train_gen = generator.flow([links[i] for i in train_idx], targets=data['labels'][train_idx], shuffle=True,
                               seed=args.seed, rank=rank, world_size=world)
    test_gen = generator.flow([links[i] for i in test_idx], targets=data['labels'][test_idx])

    graphsage = GraphSAGE(layer_sizes=args.layer_sizes, generator=generator, seed=args.seed)
    x_inp, x_out = graphsage.in_out_tensors()
    head = link_classification(edge_embedding_method="concat")(x_out, seed=args.seed)
    trainer = Trainer(graphsage, head=head, lr=args.lr, graph=graph, batch_size=args.batch_size)
    print("training path:", "static plan (HIP kernels)" if trainer.engine is not None else "autograd over the kernels")
    trainer.fit(train_gen, validation_data=test_gen if rank == 0 else None, epochs=args.epochs, verbose=rank == 0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
