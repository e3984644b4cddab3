#!/usr/bin/env python3
"""Generates tests/golden/*.json - run in the build container only (needs /root/reference).

What can be pinned against the reference's OWN code here:
``python/DataGenerator/GraphCreator.py`` and ``python/Testing/TestingNeuralNetwork.py`` import
without DGL, so their outputs on seeded inputs are recorded as fixtures:

* graphs.json    - edge lists of ``generate_graph(n, d, graph_type='reg', random_seed=s)``
                   (GraphCreator.py:31-90) for the seeds the synthetic workloads use;
* decode.json    - for seeded probability matrices: ``simple_partition_assignment``
                   (TestingNeuralNetwork.py:100-122), ``calculate_cut_value`` (:48-64),
                   ``assign_partitions`` under ``np.random.seed`` (:18-46) and
                   ``post_processing_optimization`` (:66-98).

``python/Training/TrainingNeural.py`` and ``graphExtender.py`` cannot be imported (dgl==2.0.0 is
absent and is not replaced by a stand-in), so nothing from them is recorded: the GraphConv
boundary stays "parity unpinned" and is covered by the two independent oracle formulations.
Only data (inputs / expected outputs) is written; no reference source text.
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from python.DataGenerator.GraphCreator import generate_graph  # noqa: E402
from python.Testing.TestingNeuralNetwork import (assign_partitions, calculate_cut_value,  # noqa: E402
                                                  post_processing_optimization,
                                                  simple_partition_assignment)
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    graphs = []
    for (n, d, s) in [(20, 3, 1), (50, 6, 1001), (100, 7, 1000), (64, 8, 1002), (30, 5, 1003), (200, 7, 42)]:
        g = generate_graph(n=n, d=d, graph_type="reg", random_seed=s)
        graphs.append({"n": n, "d": d, "seed": s, "nodes": list(g.nodes()),
                       "edges": [[int(u), int(v), int(w["weight"]), int(w["capacity"])]
                                 for u, v, w in g.edges(data=True)]})
    json.dump({"source": "python/DataGenerator/GraphCreator.py:generate_graph (reference, run here)",
               "networkx": __import__("networkx").__version__, "graphs": graphs},
              open(os.path.join(HERE, "graphs.json"), "w"))

    cases = []
    for (n, d, s) in [(50, 6, 1001), (100, 7, 1000), (200, 7, 42)]:
        g = generate_graph(n=n, d=d, graph_type="reg", random_seed=s)
        rng = np.random.RandomState(s)
        logits = rng.standard_normal((n, 3)).astype(np.float32) * 2.0
        e = np.exp(logits - logits.max(1, keepdims=True))
        probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
        probs[5] = [0.25, 0.5, 0.25]       # plain row
        probs[7] = [0.5, 0.5, 0.0]         # tie: first maximum must win
        probs[9] = [0.0, 0.0, 1.0]         # saturated row (neural_network_testing.ipynb:1060-1063)
        simple = simple_partition_assignment(torch.from_numpy(probs))
        np.random.seed(0)
        sample0 = assign_partitions(probs)
        sample1 = assign_partitions(probs)
        np.random.seed(0)
        best, best_cut = post_processing_optimization(torch.from_numpy(probs), g, iterations=25)
        cases.append({"n": n, "d": d, "seed": s, "probs": probs.tolist(),
                      "simple_assignment": [int(x) for x in simple],
                      "simple_cut": int(calculate_cut_value(simple, g)),
                      "np_seed": 0, "sample0": [int(x) for x in sample0], "sample1": [int(x) for x in sample1],
                      "sample0_cut": int(calculate_cut_value(sample0, g)),
                      "post_iterations": 25, "post_assignment": [int(x) for x in best],
                      "post_cut": int(best_cut)})
    json.dump({"source": "python/Testing/TestingNeuralNetwork.py (reference, run here)",
               "numpy": np.__version__,
               "note": "generated under NumPy >= 2 (NEP 50): assign_partitions keeps a float32 running sum and compares the "
                       "draw in float32 there, while the reference's pinned NumPy 1.x (envList.txt:105) sums and compares "
                       "in float64 - the semantics decode.hip and the oracle follow. The two differ only for a draw within "
                       "~6e-8 of a cumulative boundary; no draw in these cases is that close (checked by the tests).",
               "cases": cases}, open(os.path.join(HERE, "decode.json"), "w"))
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".json")])


if __name__ == "__main__":
    main()
