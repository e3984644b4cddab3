"""Offline estimate of ds_read_b128 bank-conflict cycles of the LDS gathers for an ELL table
(64-B rows: bank quarter = id & 3; lane groups of a wave = row quads {0,3,5,6},{1,2,4,7},...)."""
import sys, numpy as np
sys.path.insert(0, '.')
import importlib, networkx as nx
importlib.import_module('gcn_max_cut_amd')
graph = importlib.import_module('gcn_max_cut_amd.graph')
QUADS = [[0,3,5,6],[1,2,4,7],[8,11,13,14],[9,10,12,15]]
def sim(ell, n, W, slots=None):
    rows = (n + 15)//16*16
    t = np.empty((rows, W), dtype=np.int64); t[:n] = ell[:n]; t[n:] = t[n-1]
    t = t.reshape(rows//16, 16, W)
    cyc = 0; ideal = 0
    for qd in QUADS:
        for u in (range(W) if slots is None else slots):
            x = t[:, qd, u]
            for w in range(x.shape[0]):
                a = np.unique(x[w]); cyc += np.bincount(a & 3, minlength=4).max(); ideal += 1
    return cyc / ideal
for seed in range(3):
    g = nx.random_regular_graph(7, 1000, seed=seed)
    h = graph.from_networkx(g)
    ba = graph.BatchArrays([h])
    ell = np.asarray(ba.ell).reshape(-1, ba.ell_width).astype(np.int64)
    raw = np.full((1000, 8), 1000, np.int64)
    for r in range(1000):
        c = h.col[h.rowptr[r]:h.rowptr[r+1]]; raw[r,:len(c)] = c
    print(seed, "csr-order", round(sim(raw,1000,8),3), "arranged", round(sim(ell,1000,8),3),
          "per-slot", [round(sim(ell,1000,8,[u]),2) for u in range(8)])
