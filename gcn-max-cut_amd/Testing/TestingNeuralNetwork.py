"""TestingNeuralNetwork - the inference harness functions of
``python/Testing/TestingNeuralNetwork.py`` that drive the hot path (BASELINE configs[4]).

Same names, arguments and result dictionaries as the reference for ``assign_partitions``
(:18-46), ``calculate_cut_value`` (:48-64), ``post_processing_optimization`` (:66-98),
``simple_partition_assignment`` (:100-122), ``test_single_graph`` (:124-186) and
``test_multiple_graphs`` (:188-295).  The forward runs through ``GCNSoftmax`` (HIP), the 200
sampling iterations and their cut counts run in one launch of ``gmc_decode_sample_f32``; the
uniform draws still come from numpy's global RNG in the reference's order, so with the same
``np.random.seed`` the sampled assignments and cut values are the reference's, exactly.
The reporting / plotting half of the reference module (``analyze_results`` .. ``generate_summary_report``,
:297-638) is presentation code outside the path and is not reproduced.
"""
from __future__ import annotations

import ctypes as C
from time import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import hip
from ..graph import GraphBatch, GraphHandle, from_networkx


def assign_partitions(node_probs: np.ndarray) -> List[int]:
    """One sample of the post-processing (host form, TestingNeuralNetwork.py:18-46)."""
    out = [0, 1, 2]
    for probs in node_probs[3:]:
        r = float(np.random.rand())
        running = 0.0
        for i, p in enumerate(probs):
            # double running sum, double compare: what `cumulative_prob = 0; += np.float32` does under
            # the reference's pinned NumPy 1.x (envList.txt:105) - and what decode.hip does; written with
            # Python floats so that the installed NumPy's promotion rules do not matter
            running += float(p)
            if r < running:
                out.append(i)
                break
        else:
            out.append(len(probs) - 1)
    return out


def calculate_cut_value(partition_assignment: List[int], graph) -> int:
    """Sum of the weights of edges whose endpoints differ (TestingNeuralNetwork.py:48-64)."""
    k = len(partition_assignment)
    total = 0
    for u, v, data in graph.edges(data=True):
        if u < k and v < k and partition_assignment[u] != partition_assignment[v]:
            total += data.get('weight', 1)
    return total


def simple_partition_assignment(node_probabilities: torch.Tensor) -> List[int]:
    """argmax decode with the terminals forced to 0,1,2 (TestingNeuralNetwork.py:100-122)."""
    part = torch.argmax(node_probabilities, dim=1).cpu().numpy().tolist()
    if len(part) >= 3:
        part[0], part[1], part[2] = 0, 1, 2
    return part


def _sample_on_gpu(batch: GraphBatch, P: torch.Tensor, iterations: int):
    """Draw the uniforms in the reference's order and run the fused sampler + cut count."""
    sizes = [int(n) - 3 for n in batch.sizes]
    draws = [np.random.rand(iterations, m) for m in sizes]          # graph -> iteration -> node
    uoff = np.zeros(batch.B + 1, np.int64)
    np.cumsum([d.size for d in draws], out=uoff[1:])
    dev = batch.device
    u = torch.from_numpy(np.concatenate([d.ravel() for d in draws]) if uoff[-1] else np.zeros(1)).to(dev)
    uo = torch.from_numpy(uoff).to(dev)
    assign_all = torch.empty((iterations, batch.R), dtype=torch.int8, device=dev)
    cut_all = torch.empty((batch.B, iterations), dtype=torch.float32, device=dev)
    best_assign = torch.empty(batch.R, dtype=torch.int32, device=dev)
    best_cut = torch.empty(batch.B, dtype=torch.float32, device=dev)
    best_iter = torch.empty(batch.B, dtype=torch.int32, device=dev)
    p = hip.ptr
    rc = hip.load().gmc_decode_sample_f32(batch.ref(), p(P.contiguous()), p(u), p(uo), iterations, p(assign_all),
                                          p(cut_all), p(best_assign), p(best_cut), p(best_iter), hip.stream())
    hip.check(rc, "gmc_decode_sample_f32")
    return best_assign, best_cut, cut_all


def _as_number(x: float):
    return int(x) if float(x).is_integer() else float(x)


def post_processing_optimization(node_probabilities, graph, iterations: int = 200) -> Tuple[List[int], int]:
    """Best of ``iterations`` random samples (TestingNeuralNetwork.py:66-98), on the GPU."""
    dev = hip.require_gpu()
    probs = node_probabilities if isinstance(node_probabilities, torch.Tensor) else torch.from_numpy(np.asarray(node_probabilities))
    probs = probs.detach().to(dev, torch.float32)
    if iterations <= 0:
        return None, -float('inf')
    batch = GraphBatch([from_networkx(graph)], None, dev)
    best_assign, best_cut, _ = _sample_on_gpu(batch, probs, iterations)
    return best_assign.cpu().tolist(), _as_number(best_cut.item())


def test_single_graph(model, dgl_graph, adjacency_matrix, nx_graph, terminals: List[int],
                      post_processing_iterations: int = 200) -> Dict[str, Any]:
    """Argmax decode and post-processed decode of one graph (TestingNeuralNetwork.py:124-186)."""
    try:
        with torch.no_grad():
            node_probabilities = model(dgl_graph, adjacency_matrix)
        t0 = time()
        simple_assignment = simple_partition_assignment(node_probabilities)
        simple_cut = calculate_cut_value(simple_assignment, nx_graph)
        simple_time = time() - t0
        t0 = time()
        post_assignment, post_cut = post_processing_optimization(node_probabilities, nx_graph, post_processing_iterations)
        post_time = time() - t0
        improvement = post_cut - simple_cut
        return {
            'success': True, 'nodes': len(nx_graph.nodes()), 'edges': len(nx_graph.edges()),
            'simple_cut': simple_cut, 'simple_time': simple_time, 'simple_assignment': simple_assignment,
            'post_cut': post_cut, 'post_time': post_time, 'post_assignment': post_assignment,
            'improvement': improvement,
            'improvement_percent': (improvement / simple_cut * 100) if simple_cut > 0 else 0,
            'terminals': terminals, 'node_probabilities': node_probabilities.detach().cpu().numpy(),
        }
    except Exception as e:  # the reference reports and continues (:180-186)
        return {'success': False, 'error': str(e), 'nodes': len(nx_graph.nodes()) if nx_graph else 0,
                'edges': len(nx_graph.edges()) if nx_graph else 0}


test_single_graph.__test__ = False  # harness function, not a pytest test


def test_multiple_graphs(model, processed_graphs: Dict, graph_sizes: List[int],
                         post_processing_iterations: int = 200, verbose: bool = True) -> Tuple[List[Dict], Dict]:
    """Evaluate a dataset and bucket the results by graph size (TestingNeuralNetwork.py:188-295)."""
    if verbose:
        print("Testing neural network performance...")
        print("=" * 60)
    test_results: List[Dict] = []
    results_by_size = {size: {'simple': {'cut_values': [], 'times': []},
                              'post_processed': {'cut_values': [], 'times': []}} for size in graph_sizes}
    total = len(processed_graphs)
    if verbose:
        print(f"Sample keys from processed_graphs: {list(processed_graphs.keys())[:3]}")
    for count, (key, (g, adjacency_matrix, nx_graph, terminals)) in enumerate(processed_graphs.items(), 1):
        if isinstance(key, str):
            name = key
            try:
                size = int(name.split('_')[1][1:])
            except (IndexError, ValueError):
                size = len(nx_graph.nodes())
        else:
            name = f"graph_{key}"
            size = len(nx_graph.nodes())
            closest = min(graph_sizes, key=lambda x: abs(x - size))
            if abs(closest - size) <= 5:
                size = closest
        if verbose:
            print(f"\\nProcessing graph {count}/{total}: {name}")
            print(f"  Nodes: {len(nx_graph.nodes())}, Edges: {len(nx_graph.edges())}, Size category: {size}")
        if size not in graph_sizes:
            if verbose:
                print(f"  Skipping: graph size {size} not in test configuration")
            continue
        result = test_single_graph(model, g, adjacency_matrix, nx_graph, terminals, post_processing_iterations)
        if result['success']:
            result.update({'graph_name': name, 'graph_size': size})
            test_results.append(result)
            bucket = results_by_size[size]
            bucket['simple']['cut_values'].append(result['simple_cut'])
            bucket['simple']['times'].append(result['simple_time'])
            bucket['post_processed']['cut_values'].append(result['post_cut'])
            bucket['post_processed']['times'].append(result['post_time'])
            if verbose:
                print(f"  Simple GCN:      Cut = {result['simple_cut']}, Time = {result['simple_time']:.4f}s")
                print(f"  Post-processed:  Cut = {result['post_cut']}, Time = {result['post_time']:.4f}s")
                print(f"  Improvement:     {result['improvement']:+d} ({result['improvement_percent']:+.1f}%)")
        elif verbose:
            print(f"  ✗ Error processing graph: {result['error']}")
        if verbose and count % 10 == 0:
            print(f"\\n--- Progress: {count}/{total} ({count / total * 100:.1f}%) ---")
    if verbose:
        print(f"\\n{'=' * 60}")
        print("Neural network testing completed!")
        print(f"Successfully processed: {len(test_results)}/{total} graphs")
    return test_results, results_by_size


test_multiple_graphs.__test__ = False


def decode_dataset(model, processed_graphs: Dict, post_processing_iterations: int = 200) -> List[Dict[str, Any]]:
    """Throughput form of the same evaluation (extension): ONE batched forward and ONE sampler
    launch for the whole dataset.  Results equal ``test_multiple_graphs``'s per-graph numbers when
    the RNG state is the same (uniforms are drawn graph by graph in dataset order)."""
    items = list(processed_graphs.values())
    eng = model.engine()
    model.eval()
    handles = [it[0] for it in items]
    vals = [h.edge_values(it[1]) for h, it in zip(handles, items)]
    batch = GraphBatch(handles, vals, eng.device)
    P, S, loss = eng.forward(batch, 1.0, want_loss=True)
    best_assign, best_cut, _ = _sample_on_gpu(batch, P, post_processing_iterations)
    S_host, best_host = S.cpu().numpy(), best_assign.cpu().numpy()
    simple, post = (-loss).cpu().tolist(), best_cut.cpu().tolist()
    out = []
    for g, it in enumerate(items):
        lo, hi = int(batch.goff_host[g]), int(batch.goff_host[g + 1])
        out.append({'nodes': hi - lo, 'simple_cut': _as_number(simple[g]), 'simple_assignment': S_host[lo:hi].tolist(),
                    'post_cut': _as_number(post[g]), 'post_assignment': best_host[lo:hi].tolist(),
                    'improvement': _as_number(post[g] - simple[g])})
    return out
