"""TEST INFRASTRUCTURE ONLY - CPU oracle for the GCN max-cut hot path.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``gcn-max-cut_amd``) never imports anything under ``oracle/``
and raises when the HIP extension or a GPU is missing.

What it is: a torch-CPU restatement of the reference's training/inference
algorithm *with the reference's own structure* (dense ``[n,1000]`` feature GEMM,
per-row Python one-hot loop, dense ``[n,1000]`` loss, autograd backward,
``torch.optim.Adam``, one optimizer step per graph), following

* ``python/Training/TrainingNeural.py:69-85``   GCNSoftmax.forward
* ``python/Training/TrainingNeural.py:87-106``  terminal override + straight-through one-hot
* ``python/Training/TrainingNeural.py:137-176`` dense cut term with the hard-coded 1000 pad
* ``python/Training/TrainingNeural.py:291-309`` compute_loss (only ``C`` is used)
* ``python/Training/TrainingNeural.py:311-390`` Adam set-up + one epoch
* ``python/Training/TrainingNeural.py:392-484`` early stopping / best tracking
* ``python/Training/TrainingNeural.py:537-570`` evaluate_model
* ``python/DataGenerator/graphExtender.py:8-132`` terminal relabel + padded adjacency
* ``python/commons.py:38-77``                   dense adjacency from a networkx graph

Third-party arithmetic: ``dgl==2.0.0`` (``envList.txt:36``) ``dgl.nn.pytorch.GraphConv``
and ``dgl.from_networkx`` are NOT in ``/root/reference`` and not installed.  Their
published algorithm (norm='both', weight, bias) is restated
in :func:`graph_conv`:  ``feat*outdeg^-1/2 -> (matmul first if in>out) ->
sum over in-edges -> *indeg^-1/2 -> +bias``.

PARITY STATUS: **parity unpinned at the DGL boundary** - the reference ships no
tests, fixtures, seeds or checkpoints for this path and DGL
cannot be executed here.  What *is* pinned (tests/test_oracle_golden.py):
the graph generator, cut value, argmax decode and post-processing sampler are
checked against the reference's own importable modules
(``DataGenerator/GraphCreator.py``, ``Testing/TestingNeuralNetwork.py``) through
``tests/golden/*.json`` (made by ``tests/golden/make_golden.py``); the GraphConv
restatement is cross-checked against an independent dense float64 formula
``D^-1/2 A D^-1/2 (X W) + b``; the invariants the reference's notebooks print
(502,003 parameters, 2|E| directed edges, integer-valued loss == -cut) hold.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

PAD = 1000  # the reference hard-codes this in its loss (TrainingNeural.py:171)


# --------------------------------------------------------------------------- graphs
@dataclass
class OracleGraph:
    """What the reference gets from ``dgl.from_networkx`` (graphExtender.py:102):
    a directed edge list holding both directions of every undirected edge."""
    n: int
    src: torch.Tensor  # int64 [2|E|]
    dst: torch.Tensor  # int64 [2|E|]

    def number_of_nodes(self) -> int:
        return self.n

    def number_of_edges(self) -> int:
        return int(self.src.numel())


def graph_from_networkx(nx_graph) -> OracleGraph:
    """``dgl.from_networkx`` semantics: nodes relabelled by sorted
    order, every undirected edge becomes two directed edges."""
    order = {u: i for i, u in enumerate(sorted(nx_graph.nodes()))}
    s, d = [], []
    for u, v in nx_graph.edges():
        s += [order[u], order[v]]
        d += [order[v], order[u]]
    return OracleGraph(len(order), torch.tensor(s, dtype=torch.int64),
                       torch.tensor(d, dtype=torch.int64))


def dense_adjacency(nx_graph, pad: int = PAD, dtype=torch.float32) -> torch.Tensor:
    """commons.py:65-77 + :38-63 + graphExtender.py:28-48: ``[n, pad]`` matrix holding
    the edge ``weight`` attribute, zero elsewhere (node label == row index)."""
    n = nx_graph.number_of_nodes()
    if pad < n:
        raise ValueError("N should be greater than or equal to the original matrix size.")
    out = torch.zeros((n, pad), dtype=dtype)
    for u, v, w in nx_graph.edges(data="weight", default=1):
        out[u, v] = w
        out[v, u] = w
    return out


def swap_labels(nx_graph, mapping: Dict[int, int]) -> None:
    """graphExtender.py:8-26 - simultaneous in-place relabel via temporary labels."""
    import networkx as nx
    base = max(nx_graph.nodes) + 1
    tmp = {old: base + i for i, old in enumerate(mapping)}
    nx.relabel_nodes(nx_graph, tmp, copy=False)
    back = {v: k for k, v in mapping.items()}
    nx.relabel_nodes(nx_graph, {tmp[o]: back[o] for o in mapping}, copy=False)


def normalise_terminals(nx_graph, terminals: List[int]) -> bool:
    """graphExtender.py:72-97.  Returns False for the 'skip' branch.  Mutates both
    arguments exactly like the reference (Q11)."""
    t = terminals
    has = [k in t for k in (0, 1, 2)]
    if not any(has):
        swap_labels(nx_graph, {t[0]: 0, t[1]: 1, t[2]: 2, 0: t[0], 1: t[1], 2: t[2]})
    elif has == [False, False, True]:
        t.sort()
        swap_labels(nx_graph, {t[1]: 0, t[2]: 1, 0: t[1], 1: t[2]})
    elif has == [False, True, False]:
        t.sort()
        swap_labels(nx_graph, {t[1]: 0, t[2]: 2, 0: t[1], 2: t[2]})
    elif has == [True, False, False]:
        t.sort()
        swap_labels(nx_graph, {t[1]: 1, t[2]: 2, 1: t[1], 2: t[2]})
    else:
        return False
    return True


def make_dataset(graphs: Dict, terminals: Dict, max_nodes: int = PAD) -> Dict:
    """graphExtender.py:50-132 without batch flushing:
    ``{i: [graph, A_pad[n,max_nodes], nx_graph, [0,1,2]]}``."""
    out, i = {}, 0
    for key, g in graphs.items():
        if not normalise_terminals(g, terminals[key]):
            continue
        out[i] = [graph_from_networkx(g), dense_adjacency(g, max_nodes), g, [0, 1, 2]]
        i += 1
    return out


# --------------------------------------------------------------------------- model
def xavier_uniform(shape: Tuple[int, int], gen: torch.Generator, dtype=torch.float32):
    """``nn.init.xavier_uniform_`` bound for a ``[in, out]`` GraphConv weight."""
    bound = (6.0 / (shape[0] + shape[1])) ** 0.5
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1).mul_(bound).to(dtype)


def init_params(in_feats=1000, hidden=500, classes=3, seed=0, dtype=torch.float32):
    """GraphConv.reset_parameters: xavier-uniform weights ``[in,out]``, zero bias."""
    gen = torch.Generator().manual_seed(seed)
    return {
        "conv1.weight": xavier_uniform((in_feats, hidden), gen, dtype),
        "conv1.bias": torch.zeros(hidden, dtype=dtype),
        "conv2.weight": xavier_uniform((hidden, classes), gen, dtype),
        "conv2.bias": torch.zeros(classes, dtype=dtype),
    }


def graph_conv(g: OracleGraph, feat: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor):
    """DGL 2.0.0 ``GraphConv.forward`` with norm='both' (SURVEY App. A steps 1-6)."""
    ones = torch.ones(g.src.numel(), dtype=feat.dtype)
    out_deg = torch.zeros(g.n, dtype=feat.dtype).index_add_(0, g.src, ones)
    in_deg = torch.zeros(g.n, dtype=feat.dtype).index_add_(0, g.dst, ones)
    if bool((in_deg == 0).any()):
        raise ValueError("0-in-degree nodes (DGLError in the reference stack)")
    h = feat * out_deg.clamp(min=1).pow(-0.5).unsqueeze(1)
    if weight.shape[0] > weight.shape[1]:
        h = h @ weight
        h = torch.zeros((g.n, h.shape[1]), dtype=h.dtype).index_add(0, g.dst, h[g.src])
    else:
        h = torch.zeros((g.n, h.shape[1]), dtype=h.dtype).index_add(0, g.dst, h[g.src])
        h = h @ weight
    return h * in_deg.clamp(min=1).pow(-0.5).unsqueeze(1) + bias


def forward(params: Dict[str, torch.Tensor], g: OracleGraph, x: torch.Tensor) -> torch.Tensor:
    """TrainingNeural.py:79-85 (dropout p=0.0 is the identity)."""
    h = torch.relu(graph_conv(g, x, params["conv1.weight"], params["conv1.bias"]))
    z = graph_conv(g, h, params["conv2.weight"], params["conv2.bias"])
    return torch.softmax(z, dim=1)


def forward_dense_f64(params, nx_graph, pad: int = PAD) -> np.ndarray:
    """Independent formulation used to cross-check :func:`graph_conv`:
    ``softmax(Ahat relu(Ahat X W1 + b1) W2 + b2)`` with a dense float64 ``Ahat``."""
    n = nx_graph.number_of_nodes()
    a = np.zeros((n, n))
    for u, v in nx_graph.edges():
        a[u, v] = a[v, u] = 1.0
    dis = 1.0 / np.sqrt(np.maximum(a.sum(1), 1.0))
    ahat = dis[:, None] * a * dis[None, :]
    x = dense_adjacency(nx_graph, pad, torch.float64).numpy()
    p = {k: v.double().numpy() for k, v in params.items()}
    h = np.maximum(ahat @ (x @ p["conv1.weight"]) + p["conv1.bias"], 0.0)
    z = ahat @ (h @ p["conv2.weight"]) + p["conv2.bias"]
    e = np.exp(z - z.max(1, keepdims=True))
    return e / e.sum(1, keepdims=True)


# --------------------------------------------------------------------------- loss
def override_terminals(p: torch.Tensor) -> torch.Tensor:
    """TrainingNeural.py:87-94: rows 0,1,2 <- e0,e1,e2 (straight-through)."""
    out = p.clone()
    eye = torch.eye(3, dtype=p.dtype)
    for i in range(3):
        out[i] = eye[i] + p[i] - p[i].detach()
    return out


def straight_through_one_hot(p: torch.Tensor) -> torch.Tensor:
    """TrainingNeural.py:96-106: per-row Python loop, first max wins ties."""
    rows = []
    for i in range(p.shape[0]):
        hot = torch.zeros_like(p[i])
        hot[torch.argmax(p[i])] = 1.0
        rows.append(hot + p[i] - p[i].detach())
    return torch.stack(rows)


def cut_loss(s: torch.Tensor, a_pad: torch.Tensor, C: float = 1.0) -> torch.Tensor:
    """TrainingNeural.py:154-176 + :291-309: ``-C * sum(A_pad * (1 - pad(S S^T)))/2``."""
    n = s.shape[0]
    sst = torch.zeros((n, PAD), dtype=s.dtype)
    sst[:, :n] = s @ s.T
    return C * (-1.0 * (torch.sum(a_pad * (1 - sst)) / 2))


def graph_loss(params, g, a_pad, C: float = 1.0):
    p = forward(params, g, a_pad)
    s = straight_through_one_hot(override_terminals(p))
    return cut_loss(s, a_pad, C), p


def partition_of(p: torch.Tensor) -> np.ndarray:
    """argmax partition with terminals forced (what the loss sees)."""
    a = torch.argmax(p, dim=1).numpy().copy()
    a[:3] = [0, 1, 2]
    return a


# --------------------------------------------------------------------------- training
def loss_and_grads(params: Dict[str, torch.Tensor], items: Sequence, C: float = 1.0):
    """Sum of per-graph losses and its autograd gradient w.r.t. the four parameters
    (batch of one == one reference step, TrainingNeural.py:371-386)."""
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    total = None
    probs = []
    for (g, a_pad, _nx, _t) in items:
        l, p = graph_loss(leaf, g, a_pad, C)
        probs.append(p.detach())
        total = l if total is None else total + l
    total.backward()
    return float(total.item()), {k: v.grad.clone() for k, v in leaf.items()}, probs


class Trainer:
    """Reference-structured trainer: torch Adam over the four GraphConv tensors.
    (The reference also hands Adam an unused ``nn.Embedding`` whose grad stays None,
    TrainingNeural.py:332-337 - it never changes and is omitted here.)"""

    def __init__(self, params: Dict[str, torch.Tensor], lr: float = 1e-3, C: float = 1.0):
        self.params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr)
        self.C = C

    def step(self, items: Sequence) -> float:
        """One optimizer step on the summed loss of ``items`` (len 1 == reference)."""
        total = None
        for (g, a_pad, _nx, _t) in items:
            l, _ = graph_loss(self.params, g, a_pad, self.C)
            total = l if total is None else total + l
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        return float(total.item())

    def epoch_sequential(self, dataset: Dict) -> float:
        """TrainingNeural.py:341-390: one Adam step per graph, dict order."""
        return sum(self.step([item]) for item in dataset.values())

    def epoch_batched(self, dataset: Dict, batch: Optional[int] = None) -> float:
        items = list(dataset.values())
        batch = batch or len(items)
        return sum(self.step(items[i:i + batch]) for i in range(0, len(items), batch))

    def state(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone() for k, v in self.params.items()}


def early_stop_trace(losses: Sequence[float], tolerance: float, patience: int):
    """TrainingNeural.py:421-444 replayed on a given loss sequence.  Returns
    ``(stop_epoch or None, best_loss, loss_history)``; Q8: the break happens after the
    append and before best/prev are updated."""
    best, prev, count, hist = float("inf"), float("inf"), 0, []
    for epoch, cur in enumerate(losses):
        hist.append(cur)
        if epoch > 0 and (cur > prev or abs(prev - cur) <= tolerance):
            count += 1
            if count >= patience:
                return epoch, best, hist
        else:
            count = 0
        if cur < best:
            best = cur
        prev = cur
    return None, best, hist


def evaluate(params, dataset: Dict, C: float = 1.0) -> Dict[str, float]:
    """TrainingNeural.py:537-570."""
    tot, k = 0.0, 0
    with torch.no_grad():
        for (g, a_pad, _nx, _t) in dataset.values():
            l, _ = graph_loss(params, g, a_pad, C)
            tot += float(l.item())
            k += 1
    return {"average_loss": tot / k if k else 0, "total_loss": tot, "num_samples": k}


# --------------------------------------------------------------------------- decode (section 8f-1)
def cut_value(assign: Sequence[int], nx_graph) -> int:
    """Testing/TestingNeuralNetwork.py:48-64."""
    c = 0
    for u, v, w in nx_graph.edges(data="weight", default=1):
        if u < len(assign) and v < len(assign) and assign[u] != assign[v]:
            c += w
    return c


def sample_partition(probs: np.ndarray, rng_rand) -> List[int]:
    """Testing/TestingNeuralNetwork.py:18-46: one uniform draw per non-terminal node, compared with
    the running sum of the row, last class as fallback.  ``rng_rand`` is ``np.random.rand``-like.

    STATED SEMANTICS (they depend on the NumPy the reference runs under): the reference's pinned
    environment is NumPy 1.x (envList.txt:105), where ``cumulative_prob = 0; cumulative_prob +=
    np.float32`` promotes to float64 - the float32 probabilities are summed in DOUBLE and compared in
    double with the draw.  That is what this function (and decode.hip) does, written with explicit
    Python floats so that it does not depend on the NumPy installed here.  Under NumPy >= 2 (NEP 50)
    the reference's source would keep a float32 sum and compare in float32; the two agree unless a
    draw lies within ~6e-8 of a cumulative boundary (see :func:`sample_partition_numpy2`)."""
    out = [0, 1, 2]
    for row in probs[3:]:
        r = float(rng_rand())
        acc = 0.0
        for i, pr in enumerate(row):
            acc += float(pr)
            if r < acc:
                out.append(i)
                break
        else:
            out.append(len(row) - 1)
    return out


def sample_partition_numpy2(probs: np.ndarray, rng_rand) -> List[int]:
    """The same lines under NumPy >= 2 promotion (float32 running sum, draw rounded to float32 for the
    comparison) - only used by tests to show where the two environments part."""
    out = [0, 1, 2]
    for row in probs[3:]:
        r = np.float32(rng_rand())
        acc = np.float32(0)
        for i, pr in enumerate(row):
            acc = np.float32(acc + np.float32(pr))
            if r < acc:
                out.append(i)
                break
        else:
            out.append(len(row) - 1)
    return out


def post_process(probs: np.ndarray, nx_graph, iterations: int, rng_rand):
    """Testing/TestingNeuralNetwork.py:66-98: keep the strictly best sample."""
    best, best_cut = None, -float("inf")
    for _ in range(iterations):
        a = sample_partition(probs, rng_rand)
        c = cut_value(a, nx_graph)
        if c > best_cut:
            best, best_cut = a, c
    return best, best_cut


# --------------------------------------------------------------------------- synthetic inputs (section 8d)
def regular_graph(n: int, d: int, seed: int):
    """DataGenerator/GraphCreator.py:70-90 (graph_type='reg')."""
    import networkx as nx
    tmp = nx.convert_node_labels_to_integers(nx.random_regular_graph(d=d, n=n, seed=seed))
    g = nx.Graph()
    g.add_nodes_from(sorted(tmp.nodes()))
    g.add_edges_from(tmp.edges)
    for u, v in g.edges():
        g[u][v]["weight"] = 1
        g[u][v]["capacity"] = 1
    return g


def seeded_terminals(n: int, seed: int, k: int = 3) -> List[int]:
    """GraphCreator.py:93-109 with a seeded RNG (the reference's is unseeded)."""
    import random
    return random.Random(seed).sample(range(n), k)


def synthetic_dataset(specs: Sequence[Tuple[int, int, int]], max_nodes: int = PAD) -> Dict:
    """``specs`` = [(n, d, seed), ...] -> processed dataset dict."""
    graphs = {i: regular_graph(n, d, s) for i, (n, d, s) in enumerate(specs)}
    terms = {i: seeded_terminals(n, s) for i, (n, d, s) in enumerate(specs)}
    return make_dataset(graphs, terms, max_nodes)
