"""Shared builders for the parity tests: the same seeded graphs go through the product's
GraphExtender (-> GraphHandle + padded adjacency) and through the oracle's."""
import copy

import numpy as np
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R


def make_graphs(specs):
    graphs = {i: R.regular_graph(n, d, s) for i, (n, d, s) in enumerate(specs)}
    terms = {i: R.seeded_terminals(n, s) for i, (n, d, s) in enumerate(specs)}
    return graphs, terms


def product_dataset(specs, max_nodes=1000):
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    graphs, terms = make_graphs(specs)
    return GE.process_graphs_from_folder(graphs, terms, max_nodes)


def oracle_dataset(specs, max_nodes=1000):
    graphs, terms = make_graphs(specs)
    return R.make_dataset(graphs, terms, max_nodes)


def csrs_of(dataset):
    return [CO.csr_of(item[2]) for item in dataset.values()]


def np_params(state):
    return {k: v.detach().cpu().numpy().astype(np.float32).copy() for k, v in state.items()}


def weighted_copy(dataset_specs, seed=0):
    """Graphs with non-unit integer weights (exercises the `vals` path)."""
    graphs, terms = make_graphs(dataset_specs)
    rng = np.random.RandomState(seed)
    for g in graphs.values():
        for u, v in g.edges():
            g[u][v]["weight"] = int(rng.randint(1, 4))
    return graphs, terms
