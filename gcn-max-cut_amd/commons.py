"""commons - the I/O and adjacency helpers of ``python/commons.py`` the hot path uses.

``save_object`` / ``open_file`` (commons.py:22-36) keep their pickle format.
``gen_adj_matrix`` / ``qubo_dict_to_torch`` (commons.py:38-77) keep their signatures but
are built from the edge list instead of an O(n^2) Python dictionary fill.
The reference's star-import namespace (torch, nn, F, nx, chain, permutations, ...) that its
modules rely on via ``from python.commons import *`` is re-exported for notebook code.
"""
import os  # noqa: F401
import pickle
import random  # noqa: F401
from collections import OrderedDict, defaultdict  # noqa: F401
from itertools import chain, combinations, islice, permutations  # noqa: F401
from time import time  # noqa: F401

import networkx as nx  # noqa: F401
import numpy as np  # noqa: F401
import torch
import torch as th  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401


def save_object(obj, filename):
    with open(filename, 'wb') as outp:
        pickle.dump(obj, outp, pickle.HIGHEST_PROTOCOL)


def open_file(filename):
    """Unpickle a dataset / results file (commons.py:26-36)."""
    try:
        with open(filename, 'rb') as inp:
            return pickle.load(inp)
    except ModuleNotFoundError as exc:
        if exc.name and exc.name.split('.')[0] == 'dgl':
            raise ModuleNotFoundError(
                f"{filename} was written by the DGL-based reference and holds DGLGraph objects; "
                "re-create it with DataGenerator.graphExtender.process_graphs_from_folder "
                "(the graph handle written here needs no DGL)", name=exc.name) from exc
        raise


def gen_adj_matrix(nx_G):
    """{(u, v): weight} for every ordered node pair, 0 where there is no edge (commons.py:65-77)."""
    adj = defaultdict(int)
    nodes = list(nx_G.nodes)
    for u in nodes:
        for v in nodes:
            adj[(u, v)] = 0
    for u, v, w in nx_G.edges(data='weight'):
        adj[(u, v)] = w
        adj[(v, u)] = w
    return adj


def adjacency_tensor(nx_G, torch_dtype=None, torch_device=None):
    """Dense [n,n] weight matrix straight from the edge list (what
    ``qubo_dict_to_torch(nx_G, gen_adj_matrix(nx_G))`` produces, without the n^2 dict)."""
    n = len(nx_G.nodes)
    edges = list(nx_G.edges(data='weight'))
    mat = torch.zeros(n, n)
    if edges:
        u = torch.tensor([e[0] for e in edges], dtype=torch.long)
        v = torch.tensor([e[1] for e in edges], dtype=torch.long)
        w = torch.tensor([e[2] for e in edges], dtype=torch.float32)
        mat[u, v] = w
        mat[v, u] = w
    if torch_dtype is not None:
        mat = mat.type(torch_dtype)
    if torch_device is not None:
        mat = mat.to(torch_device)
    return mat


def qubo_dict_to_torch(nx_G, Q, torch_dtype=None, torch_device=None):
    """Dictionary {(x, y): val} -> dense tensor (commons.py:38-63)."""
    n = len(nx_G.nodes)
    mat = torch.zeros(n, n)
    if Q:
        keys = list(Q.keys())
        xs = torch.tensor([k[0] for k in keys], dtype=torch.long)
        ys = torch.tensor([k[1] for k in keys], dtype=torch.long)
        mat[xs, ys] = torch.tensor([float(Q[k]) for k in keys])
    if torch_dtype is not None:
        mat = mat.type(torch_dtype)
    if torch_device is not None:
        mat = mat.to(torch_device)
    return mat
