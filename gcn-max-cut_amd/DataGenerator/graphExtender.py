"""graphExtender - drop-in API of ``python/DataGenerator/graphExtender.py``.

Produces the dataset the training path consumes:
``{i: [graph_handle, adjacency[n, max_nodes] fp32, nx_graph, [0, 1, 2]]}``
(graphExtender.py:114).  Entry 0 is a :class:`gcn_max_cut_amd.graph.GraphHandle` (CSR,
picklable, no DGL) where the reference stores a ``DGLGraph``; entry 1 is the same dense
zero-padded adjacency the notebooks print and pass to the model.  The reference's
behaviours are kept: the caller's graphs are relabelled in place, the
terminal lists are sorted in place, graphs with two or more of {0,1,2} already terminal are
skipped, and a batch flush resets the returned dictionary.
"""
from __future__ import annotations

import traceback
from typing import Dict, List, Optional, Tuple  # noqa: F401

import networkx as nx
import torch

from ..commons import adjacency_tensor, open_file, save_object
from ..graph import from_networkx

TORCH_DEVICE = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
TORCH_DTYPE = torch.float32


def swap_graph_nodes(graph, mapping):
    """Apply a label permutation in place through temporary labels (graphExtender.py:8-26)."""
    first_free = max(graph.nodes) + 1
    parked = {label: first_free + k for k, label in enumerate(mapping)}
    nx.relabel_nodes(graph, parked, copy=False)
    target = {new: old for old, new in mapping.items()}
    nx.relabel_nodes(graph, {parked[label]: target[label] for label in mapping}, copy=False)


def extend_matrix_torch_2(matrix, N, torch_dtype=None, torch_device=None):
    """[n,n] -> [n,N], zero-padded on the right (graphExtender.py:28-48)."""
    n = matrix.shape[0]
    if N < n:
        raise ValueError("N should be greater than or equal to the original matrix size.")
    out = torch.zeros((n, N), dtype=torch_dtype or matrix.dtype, device=torch_device or matrix.device)
    out[:, :n] = matrix
    return out


def _terminal_swap(terminals: List[int]) -> Optional[Dict[int, int]]:
    """Label permutation that moves the terminals onto {0,1,2} (graphExtender.py:72-97), or
    None for the combinations the reference skips.  Sorts ``terminals`` in place exactly
    where the reference does."""
    present = tuple(k in terminals for k in (0, 1, 2))
    if present == (False, False, False):
        t = terminals
        return {t[0]: 0, t[1]: 1, t[2]: 2, 0: t[0], 1: t[1], 2: t[2]}
    slots = {(False, False, True): (0, 1), (False, True, False): (0, 2), (True, False, False): (1, 2)}
    if present not in slots:
        return None
    terminals.sort()
    a, b = slots[present]
    return {terminals[1]: a, terminals[2]: b, a: terminals[1], b: terminals[2]}


def process_graphs_from_folder(all_graphs: Dict, all_terminals: Dict, max_nodes: int,
                               save_batch_size: Optional[int] = None,
                               output_filename_prefix: str = "processed_graphs") -> Dict:
    """Normalise terminals to node ids 0,1,2 and emit the training dataset
    (graphExtender.py:50-132)."""
    datasetItem = {}
    i = 0
    skipped = 0
    filename = terminals = graph = None
    try:
        for filename, graph in all_graphs.items():
            terminals = all_terminals[filename]
            mapping = _terminal_swap(terminals)
            if mapping is None:
                skipped += 1
                continue
            swap_graph_nodes(graph, mapping)
            print(f"Terminal swapped {i}")

            handle = from_networkx(graph).to(TORCH_DEVICE)
            q_torch = adjacency_tensor(graph, torch_dtype=TORCH_DTYPE, torch_device=TORCH_DEVICE)
            full_matrix = extend_matrix_torch_2(q_torch, max_nodes, torch_dtype=TORCH_DTYPE,
                                                torch_device=TORCH_DEVICE)
            datasetItem[i] = [handle, full_matrix, graph, [0, 1, 2]]
            i += 1

            if save_batch_size and (i % save_batch_size == 0):
                batch_filename = f'{output_filename_prefix}_{i}.pkl'
                save_object(datasetItem, batch_filename)
                print(f"Saved batch to {batch_filename}")
                datasetItem = {}
            print(f"Graph finished: {i}")
    except Exception:
        print(f"Exception occurred at graph {i}, filename {filename}, terminals {terminals}")
        print(f"Graph nodes: {graph.number_of_nodes() if graph is not None else None}")
        print(traceback.format_exc())

    print(f"Skipped items: {skipped}")
    return datasetItem


def load_and_process_graphs(graphs_filename: str, terminals_filename: str, max_nodes: int,
                            output_filename: str, save_batch_size: Optional[int] = None) -> None:
    """graphExtender.py:134-161."""
    print(f"Loading graphs from {graphs_filename}")
    all_graphs = open_file(graphs_filename)
    print(f"Loading terminals from {terminals_filename}")
    all_terminals = open_file(terminals_filename)
    print(f"Processing {len(all_graphs)} graphs with max_nodes={max_nodes}")
    processed = process_graphs_from_folder(all_graphs, all_terminals, max_nodes,
                                           save_batch_size=save_batch_size,
                                           output_filename_prefix=output_filename.replace('.pkl', ''))
    if processed:
        print(f"Saving final dataset to {output_filename}")
        save_object(processed, output_filename)


def save_processed_graphs(processed_data: Dict, filename: str) -> None:
    save_object(processed_data, filename)
    print(f"Saved processed graphs to {filename}")
