"""TrainingNeural - drop-in API of ``python/Training/TrainingNeural.py`` on MI355X.

Same public names, signatures, return values, prints and checkpoint layout as the
reference module; the arithmetic of the hot loop (``train_single_epoch`` :341-390,
``GCNSoftmax.forward`` :79-85, ``evaluate_model`` :537-570) runs in the hand-written HIP
kernels of ``libgcnmaxcut_hip.so`` through :class:`gcn_max_cut_amd.engine.FusedEngine`.
There is no CPU fallback: without the HIP library and a GPU the compute entry points
raise :class:`gcn_max_cut_amd.hip.HipExtensionError`.

Deliberately kept quirks: the features are the padded adjacency (Q1),
the loss pads to a hard-coded 1000 (Q2), the "best" state aliases the live parameters
(Q4), one Adam step per graph in dataset order (Q5), early-stop bookkeeping (Q8), save
names (Q9).  Extension (keyword-only / environment, default off): ``graphs_per_step`` > 1
switches to batched steps (one Adam step per batch of graphs, summed loss), and when
``torch.distributed`` is initialised each rank trains on its shard of every batch with one
RCCL all-reduce of the flat gradient per step.
"""
from __future__ import annotations

import os
import warnings
import random  # noqa: F401  (reference namespace)
from dataclasses import dataclass
from itertools import chain, permutations
from time import perf_counter, sleep, time
from typing import Callable, Dict, List, Optional, Tuple  # noqa: F401

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

from .. import hip
from ..commons import open_file, save_object  # noqa: F401
from ..engine import PARAM_ORDER, FusedEngine, dp_active, shard_by_weight, shard_for_rank  # noqa: F401
from ..graph import GraphBatch, GraphHandle

TORCH_DEVICE = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
TORCH_DTYPE = torch.float32
_LOSS_PAD = 1000  # TrainingNeural.py:171


@dataclass
class TrainingConfig:
    """Configuration class for training parameters (TrainingNeural.py:36-67)."""
    n_nodes: int = 1000
    dim_embedding: Optional[int] = None
    hidden_dim: Optional[int] = None
    dropout: float = 0.0
    number_classes: int = 3
    learning_rate: float = 0.001
    number_epochs: int = 1000
    tolerance: float = 1e-4
    patience: int = 20
    prob_threshold: float = 0.5
    A: float = 0.0
    C: float = 1.0
    penalty: float = 1000.0
    save_directory: Optional[str] = None
    save_frequency: int = 100

    def __post_init__(self):
        if self.dim_embedding is None:
            self.dim_embedding = self.n_nodes
        if self.hidden_dim is None:
            self.hidden_dim = self.dim_embedding // 2


# --------------------------------------------------------------------------- model
class GraphConv(nn.Module):
    """Parameter container with DGL ``GraphConv``'s layout and init (weight ``[in,out]``,
    xavier-uniform; bias zeros).  The arithmetic lives in the HIP library."""

    def __init__(self, in_feats: int, out_feats: int):
        super().__init__()
        self._in_feats, self._out_feats = in_feats, out_feats
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.empty(out_feats))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        nn.init.zeros_(self.bias)

    def extra_repr(self):
        return f"in={self._in_feats}, out={self._out_feats}, normalization=both"


class _GCNForward(torch.autograd.Function):
    """softmax(conv2(relu(conv1(A)))) with a HIP backward, for callers that build their own
    loss from the probabilities (the reference's override/one-hot/compute_loss chain)."""

    @staticmethod
    def forward(ctx, net, batch, *params):
        eng = net.engine()
        ctx.dropout = eng.dropout_state()    # (p, seed) this forward runs with: the backward needs the same p
        ws = torch.empty(eng.workspace_bytes(batch, True), dtype=torch.uint8, device=eng.device)
        P, _, _ = eng.forward(batch, ws=ws)
        ctx.net, ctx.batch, ctx.ws = net, batch, ws
        ctx.save_for_backward(P)
        return P

    @staticmethod
    def backward(ctx, gp):
        (P,) = ctx.saved_tensors
        eng = ctx.net.engine()
        now = eng.dropout_state()
        eng.set_dropout(*ctx.dropout)
        try:
            g = eng.backward_from_gp(ctx.batch, P, gp.to(torch.float32), ws=ctx.ws)
            return (None, None) + tuple(g[k].clone() for k in PARAM_ORDER)
        finally:
            eng.set_dropout(*now)


class GCNSoftmax(nn.Module):
    """Graph Convolutional Network with softmax output (TrainingNeural.py:69-85)."""

    def __init__(self, in_feats: int, hidden_size: int, num_classes: int, dropout: float, device):
        super().__init__()
        self.dropout_frac = dropout
        self.conv1 = GraphConv(in_feats, hidden_size).to(device)
        self.conv2 = GraphConv(hidden_size, num_classes).to(device)
        self._engine: Optional[FusedEngine] = None

    def engine(self) -> FusedEngine:
        """The fused engine owning this model's parameters (created on first use)."""
        if self._engine is None:
            w1, w2 = self.conv1.weight, self.conv2.weight
            self._engine = FusedEngine(w1.shape[0], w1.shape[1], w2.shape[1], hip.require_gpu())
        if not self._engine.owns(self):
            self._engine.adopt(self)
        return self._engine

    def forward(self, g, inputs):
        eng = self.engine()
        try:
            batch = graph_batch_of(g, inputs, eng.device)
        except NotImplementedError:
            return self._forward_dense_features(g, inputs)
        params = [dict(self.named_parameters())[k] for k in PARAM_ORDER]
        # F.dropout(h, p=self.dropout_frac, training=self.training) (:82): a fresh mask per call in train mode;
        # the engine's dropout is 0 outside of such a call (evaluate_model, decode, the trainer's own steps)
        eng.set_dropout(self.dropout_frac if self.training else 0.0)
        try:
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                return _GCNForward.apply(self, batch, *params)
            P, _, _ = eng.forward(batch)
            return P
        finally:
            eng.set_dropout(0.0)

    def _forward_dense_features(self, g, inputs):
        if self.training and self.dropout_frac > 0.0:
            raise NotImplementedError("dropout > 0 with features that are not the padded adjacency")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(
                "gradients are implemented for the reference's usage net(g, padded_adjacency) "
                "(TrainingNeural.py:373); call under torch.no_grad() for arbitrary features")
        return _dense_forward(self, g, inputs)


def _dense_forward(net: "GCNSoftmax", g: GraphHandle, inputs: torch.Tensor) -> torch.Tensor:
    """``net(g, X)`` for features that are NOT the padded adjacency (non-zeros off the edges):
    the layer-1 feature transform is then a genuine dense GEMM (rocBLAS through torch.matmul,
    the "plain library GEMM" case), everything after it runs in the HIP kernels.  Inference
    only - no gradient is defined on this path."""
    eng = net.engine()
    lib, p = hip.load(), hip.ptr
    batch = GraphBatch([g], [None], eng.device)
    v = eng.padded_views()   # (hidden dimension padded to a multiple of 4: pad columns are zeros)
    X = inputs.detach().to(eng.device, torch.float32)
    if X.shape != (g.n, eng.N):
        raise ValueError(f"features must be [{g.n}, {eng.N}], got {tuple(X.shape)}")
    T0 = ((X * batch.dinv[:, None]) @ v["conv1.weight"]).contiguous()
    F_ = eng.Fp
    H = torch.empty((g.n, F_), dtype=torch.float32, device=eng.device)
    Z0 = torch.empty((g.n, 3), dtype=torch.float32, device=eng.device)
    P = torch.empty((g.n, 3), dtype=torch.float32, device=eng.device)
    hip.check(lib.gmc_spmm_f32(p(batch.rowptr), p(batch.gcol), None, p(batch.dinv), p(T0), F_, p(v["conv1.bias"]), 1,
                               p(H), F_, g.n, F_, 0, p(v["conv2.weight"]), p(Z0), hip.stream()), "gmc_spmm_f32")
    hip.check(lib.gmc_head_f32(batch.ref(), p(Z0), 1, p(v["conv2.bias"]), 1.0, p(P), None, None, None, None,
                               hip.stream()), "gmc_head_f32")
    return P


def graph_batch_of(g: GraphHandle, inputs, device) -> GraphBatch:
    """Single-graph device batch for ``net(g, inputs)``, cached on the handle."""
    if not isinstance(g, GraphHandle):
        raise TypeError(f"expected a GraphHandle (made by process_graphs_from_folder), got {type(g)}")
    vals = g.edge_values(inputs)
    key = ("batch", None if vals is None else id(vals), str(device))
    b = g._cache.get(key)
    if b is None:
        b = GraphBatch([g], [vals], device)
        g._cache[key] = b
    return b


# --------------------------------------------------------------------------- loss helpers (torch ops)
def override_fixed_nodes(h):
    """Rows 0,1,2 <- e0,e1,e2 with straight-through gradient (TrainingNeural.py:87-94)."""
    eye = torch.eye(3, dtype=h.dtype, device=h.device)
    head = eye + h[:3] - h[:3].detach()
    return torch.cat([head, h[3:].clone()], dim=0)


def max_to_one_hot(tensor):
    """One-hot of the first maximum, straight-through (TrainingNeural.py:96-102)."""
    hot = torch.zeros_like(tensor)
    hot[torch.argmax(tensor)] = 1.0
    return hot + tensor - tensor.detach()


def apply_max_to_one_hot(output):
    """Row-wise :func:`max_to_one_hot` without the Python loop (TrainingNeural.py:104-106)."""
    hot = F.one_hot(torch.argmax(output, dim=1), output.shape[1]).to(output.dtype)
    return hot + output - output.detach()


def extend_matrix_torch_training(matrix, N):
    size = matrix.shape[0]
    if N <= size:
        return matrix
    out = torch.zeros((N, N), dtype=matrix.dtype, device=matrix.device)
    out[:size, :size] = matrix
    return out


def extend_matrix_torch(matrix, N, torch_dtype=None, torch_device=None):
    """[n,n] -> [n,N] zero-padded (TrainingNeural.py:137-152)."""
    size = matrix.shape[0]
    if N < size:
        raise ValueError("N should be greater than or equal to the original matrix size.")
    out = torch.zeros(size, N, device=matrix.device)  # default dtype whatever the input's (as the reference)
    out[:size, :size] = matrix
    if torch_dtype is not None:
        out = out.type(torch_dtype)
    if torch_device is not None:
        out = out.to(torch_device)
    return out


def calculate_HC_vectorized(s, adjacency_matrix):
    """Total weight of cut edges, ``sum(A * (1 - pad(S S^T, 1000))) / 2`` (TrainingNeural.py:154-176)."""
    same = extend_matrix_torch(s @ s.T, _LOSS_PAD)
    return torch.sum(adjacency_matrix * (1 - same)) / 2


def compute_loss(s, adjacency_matrix, A: float = 0, C: float = 1, penalty: float = 1000):
    """``C * (-cut)``; ``A`` and ``penalty`` are accepted and unused (TrainingNeural.py:291-309)."""
    return C * (-1 * calculate_HC_vectorized(s, adjacency_matrix))


def terminal_independence_penalty(s, terminal_nodes: List[int]):
    total = 0
    for i, a in enumerate(terminal_nodes):
        for b in terminal_nodes[i + 1:]:
            total = total + torch.dot(s[a], s[b])
    return total


def find_ac_parameters(graph):
    top = max(dict(graph.degree()).values())
    return top + 1, top / 2


def generate_terminal_permutations(terminal_dict: Dict):
    keys = list(terminal_dict.keys())
    return [dict(zip(keys, perm)) for perm in permutations(terminal_dict.values())]


def calculate_all_cut_legacy(q_torch, s):
    if len(s) == 0:
        return 0
    total = 0
    for k in range(s.shape[1]):
        col = s[:, k].unsqueeze(0)
        total = total + (q_torch * (col != col.t()).float()).sum() / 2
    return total / 2


def evaluate_optimal_partitioning(net, dgl_graph, inputs, adjacency_matrix, terminal_dict: Dict):
    """TrainingNeural.py:253-289 (the permutations are generated but, as in the reference,
    never handed to the model)."""
    net.eval()
    best = float('inf')
    if dgl_graph.number_of_nodes() < 30:
        inputs = torch.ones((dgl_graph.number_of_nodes(), 30))
    with torch.no_grad():
        for _perm in generate_terminal_permutations(terminal_dict):
            probs = override_fixed_nodes(net(dgl_graph, inputs))
            value = calculate_all_cut_legacy(adjacency_matrix, (probs >= 0.5).float())
            if value < best:
                best = value
    return best


# --------------------------------------------------------------------------- training
def setup_model_and_optimizer(config: TrainingConfig):
    """(model, embedding, optimizer) - TrainingNeural.py:311-339.  The embedding is never
    used by the forward (Q1) but is part of the optimizer, the checkpoint and the return
    value, as in the reference."""
    net = GCNSoftmax(config.dim_embedding, config.hidden_dim, config.number_classes,
                     config.dropout, TORCH_DEVICE)
    net = net.type(TORCH_DTYPE).to(TORCH_DEVICE)
    embed = nn.Embedding(config.n_nodes, config.dim_embedding).type(TORCH_DTYPE).to(TORCH_DEVICE)
    optimizer = torch.optim.Adam(chain(net.parameters(), embed.parameters()), lr=config.learning_rate)
    return net, embed, optimizer


def _graphs_per_step(explicit: Optional[int]) -> int:
    if explicit is not None:
        return max(1, int(explicit))
    return max(1, int(os.environ.get("GCN_MAXCUT_GRAPHS_PER_STEP", "1")))


_NOT_LANDED = np.uint32(0x7FC0DEAD)   # quiet-NaN payload no arithmetic produces: "this loss slot has not been written"
_SPIN_BEFORE_YIELD = 2048             # ~0.5 ms of looks before the polling loops start yielding the core
_POLL_DEADLINE_S = 5.0


class FusedTrainer:
    """Device-resident state of one (model, optimizer) pair: per-step graph batches, loss
    slots, and the bridge that exposes the fused Adam moments through the torch optimizer
    (so ``optimizer.state_dict()`` has the reference's layout)."""

    def __init__(self, net: GCNSoftmax, optimizer, config: TrainingConfig, graphs_per_step: int = 1,
                 local_shard: bool = False, engine=None):
        self.net, self.optimizer, self.config = net, optimizer, config
        self.eng = engine if engine is not None else net.engine()
        self.graphs_per_step = graphs_per_step
        # local_shard: `dataset` already is this rank's shard (graphs_per_step of ITS graphs per
        # step); otherwise every rank holds the whole dataset and takes its slice of each group
        self.local_shard = local_shard
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # dp: steps run the data-parallel sequence (shard step -> all-reduce -> Adam).  More than one rank, or a
        # single rank asked to (GCN_MAXCUT_DP_SINGLE_RANK=1: RCCL and the graphs around it on a one-GPU box)
        self.dp = dp_active()
        if self.dp and hasattr(self.eng, "sync_replicas"):
            self.eng.sync_replicas(0)   # one model: rank 0's parameters / moments on every replica
        # the fused forward of this trainer's steps reads the engine's slab copy of conv1.weight (FusedEngine.ensure_slab)
        self._slab = {"slab": True} if hasattr(self.eng, "ensure_slab") else {}
        self._plan_key = None
        self._ws: Optional[torch.Tensor] = None   # scratch of this trainer's steps (captured graphs point into it)
        self._graph_key = None
        self._graph_env = None
        self._poll = os.environ.get("GCN_MAXCUT_POLL_LOSS", "1") != "0"   # watch the pinned loss slots instead of a stream sync
        self._dp_graph = None       # (per-step forward/backward hipGraphs, Adam hipGraph) of a data-parallel rank
        self._dp_graph_key = None
        self.allow_graph = True     # set False to force eager launches (per-kernel probing)
        self._graph = None          # hipGraph of one whole epoch (single GPU)
        self._graph_steps = 0
        self._batches: List[GraphBatch] = []
        self._loss_slots: Optional[torch.Tensor] = None
        self._loss_host: Optional[torch.Tensor] = None
        self._step_host: Optional[torch.Tensor] = None
        self._out = None
        self.last_enqueue_s = 0.0
        self.deadline_hits = 0      # epochs whose losses did not land within the polling deadline (see _wait_for_losses)

    def invalidate(self) -> None:
        """Forget the planned batches (and the captured hipGraphs with them): the next epoch walks the dataset
        again.  Call it after editing a dataset dict in a way :meth:`prepare` cannot see (see there)."""
        self._plan_key = None
        self._graph = None
        self._dp_graph = None

    def prepare(self, dataset: Dict) -> None:
        """Plan the device batches for ``dataset`` once and keep them while it is the same dataset.  "The same" is
        decided cheaply per epoch (the reference re-reads ``dataset.items()`` every epoch, TrainingNeural.py:371; a
        full walk per 0.2 ms step would cost more than the step): the dict object and its length, plus - for every
        item of a small dataset (<= 32 items), else for the first, the last and six evenly spaced items - the
        identity of the graph handle and of the adjacency tensor and the tensor's in-place version counter.
        Replacing or editing an item those probes miss needs :meth:`invalidate`."""
        probe = None
        if dataset:
            n_items = len(dataset)
            if n_items <= 32 or not hasattr(dataset, "__reversed__"):
                picked = list(dataset.values()) if n_items <= 32 else [next(iter(dataset.values()))]
            else:
                keys = list(dataset)   # (a list of 160 ints: ~1 us)
                picked = [dataset[keys[(n_items - 1) * j // 7]] for j in range(8)]
            probe = tuple((id(it[0]), id(it[1]), getattr(it[1], "_version", 0)) for it in picked)
        key = (id(dataset), len(dataset), self.graphs_per_step, self.world, probe)
        if key == self._plan_key:
            return
        items = list(dataset.values())
        gps, dev = self.graphs_per_step, self.eng.device
        self._batches = []
        stride = gps if self.local_shard else gps * self.world
        for start in range(0, len(items), stride):
            group = items[start:start + stride]
            # this rank's contiguous share of the group, balanced by directed edges (== by count for equal graphs)
            mine = group if self.local_shard else [group[i] for i in shard_by_weight(
                [it[0].number_of_edges() for it in group], self.rank, self.world)]
            handles = [it[0] for it in mine]
            vals = [h.edge_values(it[1]) for h, it in zip(handles, mine)]
            self._batches.append(self.eng.make_batch(handles, vals))
        rmax = max((b.R for b in self._batches), default=0)
        bmax = max((b.B for b in self._batches), default=0)
        self._out = (torch.empty((rmax, 3), dtype=torch.float32, device=dev),
                     torch.empty(rmax, dtype=torch.int32, device=dev))
        self._loss_slots = torch.zeros((len(self._batches), max(bmax, 1)), dtype=torch.float32, device=dev)
        self._step_loss = torch.zeros(len(self._batches), dtype=torch.float32, device=dev)
        # pinned landing buffer of the per-graph losses: the copy is enqueued behind the epoch's
        # kernels (inside the replayed hipGraph on one GPU), the host then waits on one event
        self._loss_host = (torch.empty_like(self._loss_slots, device="cpu").pin_memory()
                           if dev.type == "cuda" else None)
        self._loss_host_np = self._loss_host.numpy() if self._loss_host is not None else None   # (shares the pinned memory)
        self._loss_host_bits = self._loss_host_np.view(np.uint32) if self._loss_host_np is not None else None
        # device-side address of that pinned buffer: the captured steps store their per-graph losses straight
        # into it (one system-scope store each, as soon as the value is final), so the host has a step's loss
        # while its backward is still running and no copy node trails the graph (GCN_MAXCUT_LOSS_ZEROCOPY=0: copy)
        self._loss_host_dev = None
        if (self._loss_host is not None and hasattr(self.eng, "lib") and self._poll
                and os.environ.get("GCN_MAXCUT_LOSS_ZEROCOPY", "1") != "0"):
            self._loss_host_dev = hip.mapped_ptr(self._loss_host)
        self._step_host = (torch.empty_like(self._step_loss, device="cpu").pin_memory()
                           if dev.type == "cuda" else None)
        self._step_host_np = self._step_host.numpy() if self._step_host is not None else None
        self._step_host_bits = self._step_host_np.view(np.uint32) if self._step_host_np is not None else None
        # data-parallel steps: the all-reduced loss of a step (the gradient's tail slot) is published to this pinned
        # buffer by a one-wave launch BEFORE the step's Adam launches, so the host has it while Adam still runs
        self._step_host_dev = None
        if (self._step_host is not None and hasattr(self.eng, "publish") and self._poll
                and os.environ.get("GCN_MAXCUT_LOSS_ZEROCOPY", "1") != "0"):
            self._step_host_dev = hip.mapped_ptr(self._step_host)
        # private scratch, sized for the largest step: the engine's own scratch is re-allocated whenever a
        # later call (evaluate_model on a bigger batch, another trainer) needs more, which would leave a
        # captured hipGraph replaying into freed memory
        if hasattr(self.eng, "workspace_bytes") and self._batches:
            drop = float(getattr(self.net, "dropout_frac", 0.0) or 0.0)
            if drop > 0.0:   # the dropout sequence needs a little more scratch: size for it
                self.eng.set_dropout(drop, 0)
            need = max((self.eng.workspace_bytes(b, True) for b in self._batches if b.B), default=0)
            if drop > 0.0:
                self.eng.set_dropout(0.0)
            if need and (self._ws is None or self._ws.numel() < need):
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self._plan_key = key
        self._graph = None
        self._dp_graph = None

    def _hyper(self):
        """(lr, betas, eps) of the step: the optimizer's first param group, as ``optimizer.step()`` would use
        (TrainingNeural.py:337,386); the config's learning rate when the optimizer carries none."""
        groups = getattr(self.optimizer, "param_groups", None)
        if groups:
            g = groups[0]
            return float(g.get("lr", self.config.learning_rate)), tuple(g.get("betas", (0.9, 0.999))), float(g.get("eps", 1e-8))
        return float(self.config.learning_rate), (0.9, 0.999), 1e-8

    def _dropout(self) -> float:
        """p of F.dropout for this epoch's steps (TrainingNeural.py:82): the model's, in train mode."""
        return float(getattr(self.net, "dropout_frac", 0.0)) if getattr(self.net, "training", False) else 0.0

    def epoch(self, dataset: Dict) -> float:
        """One pass over the dataset; returns the cumulative loss (one host sync)."""
        t_entry = perf_counter()
        self.prepare(dataset)
        eng, cfg = self.eng, self.config
        drop = self._dropout()
        if drop > 0.0:
            # dropout: eager launches of the one-kernel-per-operation sequence, a fresh mask per step (the
            # captured graphs would replay one mask; every reference configuration trains with p = 0)
            lr, betas, eps = self._hyper()
            tail = eng.grad[eng.count:eng.count + 1]
            for i, batch in enumerate(self._batches):
                eng.set_dropout(drop)
                if batch.B == 0:
                    eng.grad[:eng.count + 1].zero_()
                else:
                    eng.train_fwd_bwd(batch, cfg.C, out=(self._out[0], self._out[1], self._loss_slots[i]), ws=self._ws)
                if self.dp:
                    eng.allreduce_grad()
                self._step_loss[i:i + 1].copy_(tail)
                eng.adam_step(lr, betas, eps)
            eng.set_dropout(0.0)
            return float(self._step_loss.cpu().numpy().sum(dtype=np.float64))
        use_graph = self._use_graph()
        publish = None
        # eager launches with the losses stored straight into the pinned buffer: the host is back as soon as the
        # loss kernels have run and queues the next step's launches behind this step's backward
        eager_direct = (not use_graph and not self.dp and hasattr(eng, "train_step") and self._poll
                        and getattr(self, "_loss_host_dev", None) is not None)
        poll = (use_graph or eager_direct) and self._poll and self._loss_host_np is not None
        if poll:
            self._loss_host_bits.fill(_NOT_LANDED)   # sentinel bit pattern: no kernel produces it, a NaN LOSS is not it
        if use_graph:
            self._replay_epoch()
        elif not self.dp and hasattr(eng, "train_step"):
            eng.sync_step_dev()
            self._enqueue_epoch(with_readback=eager_direct)
        else:
            # the step's loss rides in the gradient all-reduce: train_fwd_bwd leaves the shard's loss sum
            # in the slot after the gradient (GMC_MODEL_GRAD_TAIL), the all-reduce makes it the batch's
            tail = eng.grad[eng.count:eng.count + 1]
            last = len(self._batches) - 1
            lr, betas, eps = self._hyper()
            graphs = self._dp_graphs() if self._use_dp_graph() else None
            if graphs is not None:
                eng.sync_step_dev()
                if self._slab:
                    eng.ensure_slab()   # (a launch only when torch wrote the parameters since the last step)
            publish = self._step_host_dev if self.dp else None
            if publish:
                self._step_host_bits.fill(_NOT_LANDED)   # sentinel bit pattern (a NaN loss is a landed value)
            for i, batch in enumerate(self._batches):
                if batch.B == 0:
                    # this rank's shard of the step is empty (last group smaller than the world): it
                    # contributes a zero gradient and a ZERO loss - the tail slot still holds the previous
                    # step's all-reduced loss and would otherwise be added once more per empty rank
                    eng.grad[:eng.count + 1].zero_()
                elif graphs is not None:
                    graphs[0][i].replay()              # forward + loss + backward + gradient fold of my shard
                else:
                    eng.train_fwd_bwd(batch, cfg.C, out=(self._out[0], self._out[1], self._loss_slots[i]),
                                      **({"ws": self._ws} if self._ws is not None else {}), **self._slab)
                fused_publish = None
                if self.dp:
                    eng.allreduce_grad()               # ONE RCCL all-reduce of [gradient | loss] per step, eager
                    if publish and graphs is None and self._slab and hasattr(eng, "adam_step_dev"):
                        fused_publish = (tail, publish + 4 * i)   # rides in front of the Adam launch below
                    elif publish:                      # the step's loss -> pinned host slot i, ahead of Adam
                        eng.publish(tail, publish + 4 * i)
                    elif i != last:                    # the last step's slot is read in place below
                        self._step_loss[i:i + 1].copy_(tail)
                if graphs is not None:
                    graphs[1].replay()                 # Adam, step number read from / advanced in device memory
                    eng.step_count += 1
                    eng._dev_step += 1
                elif self._slab and hasattr(eng, "adam_step_dev"):
                    eng.sync_step_dev()                # (a launch only after host-stepped updates)
                    # keeps the slab copy of W1 current; with `publish`: loss store + counter tick + Adam in two launches
                    eng.adam_step_dev(lr, betas, eps, **self._slab, **({"publish": fused_publish} if fused_publish else {}))
                else:
                    eng.adam_step(lr, betas, eps)
        self.last_enqueue_s = perf_counter() - t_entry   # host time to queue the epoch's launches (bench.py reports it)
        if self.dp:   # one host sync per epoch
            if not self._batches:
                return 0.0
            if self._step_host is not None:
                host, bits = self._step_host_np, self._step_host_bits
                if self._poll and not publish:
                    bits.fill(_NOT_LANDED)   # sentinel bit pattern (a NaN loss is a landed value)
                if publish:
                    pass            # every step has already sent its loss
                elif last == 0:   # one step per epoch: its loss goes from the gradient's tail slot to the host
                    self._step_host.copy_(tail, non_blocking=True)
                else:
                    self._step_loss[last:last + 1].copy_(tail)
                    self._step_host.copy_(self._step_loss, non_blocking=True)
                if self._poll:   # watch the pinned slots instead of sleeping in the stream sync (see _wait_for_losses)
                    spins, deadline = 0, None
                    while True:
                        if bits[last] != _NOT_LANDED and not (bits == _NOT_LANDED).any():
                            return float(host.sum(dtype=np.float64))
                        spins += 1
                        if spins > _SPIN_BEFORE_YIELD and spins & 63 == 0:
                            sleep(0)   # give the core away: RCCL's proxy threads and the other ranks' hosts share it
                            now = time()
                            deadline = deadline or now + _POLL_DEADLINE_S
                            if now > deadline:
                                self._deadline_hit()
                                break
                torch.cuda.current_stream().synchronize()
                return float(host.sum(dtype=np.float64))
            self._step_loss[last:last + 1].copy_(tail)
            return float(sum(self._step_loss.cpu().tolist()))
        # one device->host copy per epoch; the reference adds one float per optimizer step
        # (loss.item(), :388), each the sum of that step's per-graph losses
        if self._loss_host is not None:
            if not use_graph and not eager_direct:   # (the replayed graph ends with this copy / stores directly)
                self._loss_host.copy_(self._loss_slots, non_blocking=True)
            host = self._loss_host_np
            if poll:
                return self._wait_for_losses(host)
            torch.cuda.current_stream().synchronize()
        else:
            host = self._loss_slots.cpu().numpy()
        if len(self._batches) == 1:
            return float(host[0, :self._batches[0].B].sum(dtype=np.float32))
        total = 0.0
        for i, batch in enumerate(self._batches):
            total += float(host[i, :batch.B].sum(dtype=np.float32))
        return total

    def _wait_for_losses(self, host: np.ndarray) -> float:
        """The loss kernels store every graph's loss straight into pinned host memory (or the replayed graph ends
        with that copy): the host watches that memory instead of sleeping in hipStreamSynchronize (whose wake-up
        costs ~10 us per step of a 0.25 ms step).  "Not landed yet" is a BIT PATTERN (`_NOT_LANDED`, a NaN payload
        no kernel produces) the slots are filled with before the launch, so a loss that genuinely IS NaN (diverged
        weights) counts as landed and comes back as NaN at once - as `loss.item()` would (TrainingNeural.py:387-388).
        After ~0.5 ms of spinning the loop yields the core between looks; slots that have not landed within
        `_POLL_DEADLINE_S` hand over to the stream synchronisation - which reports whatever went wrong on the device -
        and polling is switched off for the rest of the run with ONE warning (e.g. pinned memory that is not
        host-coherent: every epoch would otherwise pay the deadline).  Returns the epoch's cumulative loss."""
        bits = self._loss_host_bits
        total, deadline = 0.0, None
        for i, batch in enumerate(self._batches):
            nb = batch.B
            if nb == 0:
                continue
            row, brow, spins = host[i], bits[i], 0
            while True:
                if brow[nb - 1] != _NOT_LANDED and brow[0] != _NOT_LANDED and not (brow[:nb] == _NOT_LANDED).any():
                    total += float(row[:nb].sum(dtype=np.float32))
                    break
                spins += 1
                if spins > _SPIN_BEFORE_YIELD and spins & 63 == 0:
                    sleep(0)
                    now = time()
                    deadline = deadline or now + _POLL_DEADLINE_S
                    if now > deadline:   # the stream is done after this: whatever the slots hold IS the result
                        torch.cuda.current_stream().synchronize()   # (raises if the device faulted)
                        self._deadline_hit()
                        total += float(row[:nb].sum(dtype=np.float32))
                        break
        return total

    def _deadline_hit(self) -> None:
        self.deadline_hits += 1
        if self._poll:
            self._poll = False
            warnings.warn("GCN max-cut: the step's losses did not reach the pinned host buffer within "
                          f"{_POLL_DEADLINE_S:.0f} s of polling; falling back to stream synchronisation for the rest "
                          "of this run (is the pinned memory host-coherent? HIP_HOST_COHERENT=0 breaks zero-copy stores)")

    def _use_graph(self) -> bool:
        """On one GPU an epoch's launches (the reference's one Adam step per graph: hundreds of
        ~10 us kernels; or one batched step: six) are captured once into a hipGraph and replayed per
        epoch, which removes the per-launch host cost."""
        if self._graph_env is None:   # process-wide facts, looked up once
            self._graph_env = torch.cuda.is_available() and os.environ.get("GCN_MAXCUT_HIPGRAPH", "1") != "0"
        if not (self.allow_graph and not self.dp and self._graph_env and len(self._batches) >= 1
                and hasattr(self.eng, "train_step")):
            return False
        # ONE step per epoch (the batched schedule): four eager launches, queued behind the previous step's
        # backward as soon as its losses have reached the pinned buffer, run back to back; a graph replay per step
        # pays the ~10 us between two graph executions instead (measured: 0.2259 vs 0.2288 ms per step)
        if (len(self._batches) == 1 and self._poll and getattr(self, "_loss_host_dev", None) is not None
                and os.environ.get("GCN_MAXCUT_EAGER_SINGLE_STEP", "1") != "0"):
            return False
        return True

    def _use_dp_graph(self) -> bool:
        """Data-parallel ranks launch eagerly by default: the host has a step's loss before its Adam launches run
        (publish) and queues the next step's launches behind them, which is as fast as replaying graphs (0.246 against
        0.249 ms per step on one rank over RCCL) and keeps stream capture away from RCCL's threads.
        GCN_MAXCUT_DP_GRAPHS=1: the launches on either side of the all-reduce are replayed from hipGraphs - one
        per step for forward/loss/backward, one for Adam."""
        return (self.allow_graph and self.dp and hasattr(self.eng, "adam_step_dev") and self._ws is not None
                and torch.cuda.is_available() and os.environ.get("GCN_MAXCUT_HIPGRAPH", "1") != "0"
                and os.environ.get("GCN_MAXCUT_DP_GRAPHS", "0") == "1")

    def _dp_graphs(self):
        eng, cfg = self.eng, self.config
        hyper = self._hyper()
        key = (hyper, float(cfg.C), self._ws.data_ptr())
        if self._dp_graph is not None and self._dp_graph_key == key:
            return self._dp_graph
        lr, betas, eps = hyper
        fb = []
        for i, batch in enumerate(self._batches):
            if batch.B == 0:
                fb.append(None)
                continue
            out = (self._out[0], self._out[1], self._loss_slots[i])
            eng.train_fwd_bwd(batch, cfg.C, out=out, ws=self._ws, **self._slab)     # eager once: warms the kernels
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.train_fwd_bwd(batch, cfg.C, out=out, ws=self._ws, **self._slab)
            fb.append(g)
        before, flat, m, v = eng.step_count, eng.flat.clone(), eng.m.clone(), eng.v.clone()
        eng.sync_step_dev()
        eng.adam_step_dev(lr, betas, eps, **self._slab)                # eager once (state restored below)
        ga = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            eng.adam_step_dev(lr, betas, eps, **self._slab)
        eng.flat.copy_(flat); eng.m.copy_(m); eng.v.copy_(v)
        eng.step_count = before
        eng._dev_step = -1                                             # (the capture pass counted on the host only)
        eng.sync_step_dev()
        self._dp_graph, self._dp_graph_key = (fb, ga), key
        return self._dp_graph

    def _enqueue_epoch(self, with_readback: bool = False) -> None:
        eng, cfg = self.eng, self.config
        lr, betas, eps = self._hyper()
        direct = self._loss_host_dev if with_readback else None
        row_bytes = self._loss_slots.shape[1] * 4 if self._loss_slots is not None else 0
        for i, batch in enumerate(self._batches):
            extra = {"loss_ptr": direct + i * row_bytes} if direct else {}
            eng.train_step(batch, lr, cfg.C, out=(self._out[0], self._out[1], self._loss_slots[i]), betas=betas, eps=eps,
                           ws=self._ws, **self._slab, **extra)
        if with_readback and self._loss_host is not None and not direct:
            self._loss_host.copy_(self._loss_slots, non_blocking=True)

    def _replay_epoch(self) -> None:
        eng = self.eng
        eng.sync_step_dev()   # (no launch while this trainer's replays are the only thing stepping the optimizer)
        # the captured launches carry lr / betas / eps / C and the scratch pointer as kernel arguments
        key = (self._hyper(), float(self.config.C), self._ws.data_ptr() if self._ws is not None else 0)
        if self._graph is not None and key != self._graph_key:
            self._graph = None
        self._graph_key = key
        if self._graph is None:
            self._enqueue_epoch()                 # eager epoch: sizes the workspace, warms the kernels
            graph = torch.cuda.CUDAGraph()
            before = eng.step_count
            with torch.cuda.graph(graph):
                self._enqueue_epoch(with_readback=True)
            eng.step_count = eng._dev_step = before   # capture enqueued nothing
            self._graph, self._graph_steps = graph, len(self._batches)
            self._loss_host.copy_(self._loss_slots, non_blocking=True)   # this (eager) epoch's losses
            return
        if self._slab:
            eng.ensure_slab()   # (a launch only when torch wrote the parameters since the last replay)
        self._graph.replay()
        eng.step_count += self._graph_steps
        eng._dev_step += self._graph_steps

    def sync_optimizer_state(self) -> None:
        """Expose step / exp_avg / exp_avg_sq of the fused Adam through ``optimizer.state``."""
        if self.eng.step_count == 0:
            return
        named = dict(self.net.named_parameters())
        mv, vv = self.eng.views(self.eng.m), self.eng.views(self.eng.v)
        for k in PARAM_ORDER:
            self.optimizer.state[named[k]] = {
                'step': torch.tensor(float(self.eng.step_count)),
                'exp_avg': mv[k], 'exp_avg_sq': vv[k]}


def _trainer_for(net, optimizer, config, graphs_per_step: Optional[int] = None) -> FusedTrainer:
    gps = _graphs_per_step(graphs_per_step)
    tr = getattr(net, "_fused_trainer", None)
    if tr is None or tr.optimizer is not optimizer or tr.graphs_per_step != gps:
        tr = FusedTrainer(net, optimizer, config, gps)
        net._fused_trainer = tr
    tr.config = config
    return tr


def train_single_epoch(dataset: Dict, net, optimizer, embed, config: TrainingConfig,
                       dataset_files: Optional[List[str]] = None, *,
                       graphs_per_step: Optional[int] = None) -> float:
    """One epoch, cumulative loss (TrainingNeural.py:341-390).  The device batches planned for ``dataset`` are
    kept from epoch to epoch while it looks unchanged (:meth:`FusedTrainer.prepare` says how that is decided);
    after editing a large dataset dict in place call ``net._fused_trainer.invalidate()``."""
    net.train()
    trainer = _trainer_for(net, optimizer, config, graphs_per_step)
    if dataset_files is None:
        dataset_files = ['./nx_test_generated_graph_n200_300_d8_12_t500.pkl']
    cumulative_loss = 0.0
    for dataset_file in dataset_files:
        current = dataset if isinstance(dataset, dict) else open_file(dataset_file)
        cumulative_loss += trainer.epoch(current)
    return cumulative_loss


def _rank0() -> bool:
    """Data-parallel runs: every rank holds the same model, rank 0 alone prints and writes checkpoints
    (N processes writing ./epoch_*.pth at once would race on the same files)."""
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _checkpoint(net, optimizer, embed, epoch, loss_history, config) -> Dict:
    tr = getattr(net, "_fused_trainer", None)
    if tr is not None:
        tr.sync_optimizer_state()
    return {'epoch': epoch, 'model': net.state_dict(), 'optimizer': optimizer.state_dict(),
            'loss_history': loss_history, 'inputs': embed.weight, 'config': config}


def train_model(dataset: Dict, config: TrainingConfig, dataset_files: Optional[List[str]] = None, *,
                graphs_per_step: Optional[int] = None) -> Tuple:
    """Main training function (TrainingNeural.py:392-484):
    returns ``(model, best_loss, final_epoch, embedding_weights, loss_history)``."""
    say = print if _rank0() else (lambda *a, **k: None)
    say(f"Starting training with {config.number_epochs} epochs")
    say(f"Model: {config.n_nodes} nodes, {config.number_classes} classes")
    say(f"Device: {TORCH_DEVICE}")

    net, embed, optimizer = setup_model_and_optimizer(config)
    best_loss, best_model_state = float('inf'), None
    loss_history: List[float] = []
    patience_counter, prev_loss = 0, float('inf')
    start_time = time()
    epoch = -1

    for epoch in range(config.number_epochs):
        cumulative_loss = train_single_epoch(dataset, net, optimizer, embed, config, dataset_files,
                                             graphs_per_step=graphs_per_step)
        loss_history.append(cumulative_loss)

        stalled = cumulative_loss > prev_loss or abs(prev_loss - cumulative_loss) <= config.tolerance
        if epoch > 0 and stalled:
            patience_counter += 1
            if patience_counter >= config.patience:
                say(f'Early stopping at epoch {epoch}')
                break
        else:
            patience_counter = 0

        if cumulative_loss < best_loss:
            best_loss = cumulative_loss
            best_model_state = net.state_dict()  # aliases the live parameters (Q4)
        prev_loss = cumulative_loss

        if epoch % config.save_frequency == 0:
            say(f'Epoch: {epoch}, Cumulative Loss: {cumulative_loss:.6f}')
            if config.save_directory and _rank0():
                torch.save(_checkpoint(net, optimizer, embed, epoch, loss_history, config),
                           f'./epoch_{epoch}_loss_{cumulative_loss:.4f}_{config.save_directory}')

    if best_model_state is not None:
        net.load_state_dict(best_model_state)

    say(f'Training completed in {time() - start_time:.2f} seconds')
    say(f'Best loss: {best_loss:.6f}')

    if config.save_directory and _rank0():
        final_filename = f'./final_{config.save_directory}'
        torch.save(_checkpoint(net, optimizer, embed, epoch, loss_history, config), final_filename)
        say(f'Final model saved to {final_filename}')

    return net, best_loss, epoch, embed.weight, loss_history


def train_from_pickle(dataset_filename: str, model_name: str, n_nodes: int = 1000, **kwargs) -> Tuple:
    """Train from a dataset pickle (TrainingNeural.py:486-513)."""
    graphs_per_step = kwargs.pop('graphs_per_step', None)
    config = TrainingConfig(**{'n_nodes': n_nodes, 'save_directory': f'{model_name}.pth', **kwargs})
    print(f"Loading dataset from {dataset_filename}")
    dataset = open_file(dataset_filename)
    return train_model(dataset, config, graphs_per_step=graphs_per_step)


def train_multi_class(dataset_filename: str, model_name: str, num_classes: int = 3, **kwargs) -> Tuple:
    """TrainingNeural.py:515-535."""
    params = {'number_classes': num_classes, 'save_directory': f'{model_name}.pth', **kwargs}
    return train_from_pickle(dataset_filename, model_name, **params)


def evaluate_model(model, dataset: Dict, config: TrainingConfig) -> Dict:
    """Average / total loss over a dataset (TrainingNeural.py:537-570): forward, terminal
    override, argmax decode and cut loss for all graphs in one fused launch sequence."""
    model.eval()
    items = list(dataset.values())
    if not items:
        return {'average_loss': 0, 'total_loss': 0.0, 'num_samples': 0}
    eng = model.engine()
    handles = [it[0] for it in items]
    vals = [h.edge_values(it[1]) for h, it in zip(handles, items)]
    batch = GraphBatch(handles, vals, eng.device)
    _, _, loss = eng.forward(batch, config.C, want_loss=True)
    total = 0.0
    for value in loss.cpu().tolist():
        total += value
    return {'average_loss': total / len(items), 'total_loss': total, 'num_samples': len(items)}


def load_neural_model(model_path: str, config: TrainingConfig):
    """(model, inputs, loaded_config) from a checkpoint (TrainingNeural.py:572-609)."""
    import torch.serialization
    try:
        torch.serialization.add_safe_globals([TrainingConfig])
        checkpoint = torch.load(model_path, map_location=TORCH_DEVICE)
    except Exception:
        try:
            checkpoint = torch.load(model_path, map_location=TORCH_DEVICE, weights_only=False)
        except Exception:
            with torch.serialization.safe_globals([TrainingConfig]):
                checkpoint = torch.load(model_path, map_location=TORCH_DEVICE)
    net, embed, _ = setup_model_and_optimizer(config)
    net.load_state_dict(checkpoint['model'])
    return net, checkpoint.get('inputs', embed.weight), checkpoint.get('config', config)


def save_neural_model(model, optimizer, embed, epoch: int, loss_history: List,
                      config: TrainingConfig, model_path: str):
    """TrainingNeural.py:611-634."""
    torch.save(_checkpoint(model, optimizer, embed, epoch, loss_history, config), model_path)
    print(f'Model saved to {model_path}')


# --------------------------------------------------------------------------- legacy wrappers (:636-733)
def get_gnn_legacy(n_nodes: int, gnn_hypers: Dict, opt_params: Dict, torch_device, torch_dtype):
    return setup_model_and_optimizer(TrainingConfig(
        n_nodes=n_nodes, dim_embedding=gnn_hypers['dim_embedding'], hidden_dim=gnn_hypers['hidden_dim'],
        dropout=gnn_hypers['dropout'], number_classes=gnn_hypers['number_classes'],
        learning_rate=opt_params['lr']))


def hyperparameters_legacy(n: int = 80, d: int = 3, p=None, graph_type: str = 'reg',
                           number_epochs: int = int(1e5), learning_rate: float = 1e-4,
                           prob_threshold: float = 0.5, tol: float = 1e-4, patience: int = 100):
    dim_embedding = n
    return (n, d, p, graph_type, number_epochs, learning_rate, prob_threshold, tol, patience,
            dim_embedding, int(dim_embedding / 2))


def train_legacy_wrapper(model_name: str, filename: str = './testData/nx_generated_graph_n80_d3_t200.pkl',
                         n: int = 80):
    return train_from_pickle(filename, model_name, n_nodes=n, learning_rate=0.001, patience=20)


def train_2way_neural_legacy(model_name: str, filename: str = './testData/prepareDS.pkl'):
    return train_multi_class(filename, model_name, num_classes=2, n_nodes=4096,
                             learning_rate=0.001, patience=20, number_epochs=500)


get_gnn = get_gnn_legacy
hyperParameters = hyperparameters_legacy
train1 = train_legacy_wrapper
train_2wayNeural = train_2way_neural_legacy
FIndAC = find_ac_parameters
GetOptimalNetValue = evaluate_optimal_partitioning
calculateAllCut = calculate_all_cut_legacy
LoadNeuralModel = load_neural_model
