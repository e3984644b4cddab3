"""Fused HIP engine: forward / training step / Adam over the C-ABI library.

Holds the model's four GraphConv tensors as views of ONE flat fp32 buffer
``[W1 | b1 | W2 | b2]`` (so a single fused Adam sweep and a single RCCL all-reduce cover
them), the gradient / Adam-moment buffers of the same shape, and the scratch workspace.
All kernels are enqueued on torch's current HIP stream through ``gcnmaxcut.h``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import hip
from .graph import GraphBatch

PARAM_ORDER = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias")


def flat_layout(N: int, F: int, K: int) -> Tuple[List[int], int]:
    sizes = [N * F, F, F * K, K]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    return offs, offs[-1]


class FusedEngine:
    """One per model.  ``adopt`` re-homes a module's parameters into the flat buffer."""

    def __init__(self, N: int, F: int, K: int, device: Optional[torch.device] = None):
        self.device = device or hip.require_gpu()
        self.lib = hip.load()
        if K != 3:
            raise ValueError("number_classes must be 3: the terminal override is 3-wide "
                             "(TrainingNeural.py:91-93)")
        if F < 1 or F > 1024:
            raise ValueError("hidden_dim must be in 1..1024 on this path")
        self.N, self.F, self.K = N, F, K
        # Any hidden_dim (TrainingNeural.py:42,66-67 accept any int; n_nodes=50 gives 25): the kernels work on 16-byte
        # column groups, so the flat buffer carries the hidden dimension padded to a multiple of 4 (Fp).  Pad columns of
        # W1 / entries of b1 / rows of W2 are 0 and stay 0 (their activations are relu(0), every gradient entry is an
        # exact 0, Adam of 0 is 0); the module's parameters and state_dict keep the logical shapes as views [:, :F].
        self.Fp = (F + 3) // 4 * 4
        self.offs, self.count = flat_layout(N, self.Fp, K)
        # + 4 floats of tail: slot `count` carries the summed loss through the all-reduce
        self.flat = torch.zeros(self.count + 4, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)  # device mirror of step_count
        self._dev_step = 0   # what step_dev holds (host shadow): replayed / device-stepped launches advance both
        self._ws: Optional[torch.Tensor] = None
        self._model = hip.GmcModel()
        self._refresh_model()
        # slab copy of conv1.weight for the fused forward (gmc_model.W1_slab): used by the calls that ask for it
        # (FusedTrainer's steps), kept current by the fused Adam kernels, re-built when torch wrote the parameters
        self.w1_slab: Optional[torch.Tensor] = None
        self._slab_sig = None
        self._adopted: List = []
        self.slab_enabled = hip.HAS_SLAB and os.environ.get("GCN_MAXCUT_W1_SLAB", "1") != "0"

    # ---- parameters
    def views(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """The four tensors in the reference's (logical) shapes: views of ``buf`` (for hidden_dim % 4 != 0
        ``conv1.weight`` is a strided view of the padded rows)."""
        p, F = self.padded_views(buf), self.F
        return {"conv1.weight": p["conv1.weight"][:, :F], "conv1.bias": p["conv1.bias"][:F],
                "conv2.weight": p["conv2.weight"][:F], "conv2.bias": p["conv2.bias"]}

    def padded_views(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """The same four tensors as the kernels see them: hidden dimension padded to Fp, contiguous."""
        buf = self.flat if buf is None else buf
        o, N, F, K = self.offs, self.N, self.Fp, self.K
        return {"conv1.weight": buf[o[0]:o[1]].view(N, F), "conv1.bias": buf[o[1]:o[2]],
                "conv2.weight": buf[o[2]:o[3]].view(F, K), "conv2.bias": buf[o[3]:o[4]]}

    def _refresh_model(self) -> None:
        v = self.padded_views()
        self._model = hip.GmcModel(N=self.N, F=self.Fp, K=self.K, flags=hip.MODEL_GRAD_TAIL,
                                   W1=hip.ptr(v["conv1.weight"]), b1=hip.ptr(v["conv1.bias"]),
                                   W2=hip.ptr(v["conv2.weight"]), b2=hip.ptr(v["conv2.bias"]))

    # ---- slab copy of W1
    def _param_signature(self):
        # every torch-visible in-place write bumps the version counter of the tensor it went through: the flat
        # buffer (and its views) or an adopted nn.Parameter.  The library's kernels write through raw pointers
        # and bump nothing - they keep the slab current themselves.
        return (self.flat._version,) + tuple(p._version for p in self._adopted)

    def ensure_slab(self) -> int:
        """Device pointer of the slab copy of conv1.weight (gmc_model.W1_slab), re-built first when a torch
        operation wrote the parameters since it was last known to be current (optimizer.step(), load_state_dict,
        a broadcast, ...).  0 when the copy is switched off (GCN_MAXCUT_W1_SLAB=0 / an older library).  A write
        torch cannot see (``param.data`` arithmetic, foreign kernels) is picked up one step late: the Adam
        kernels refresh the copy from the row-major weights on every step."""
        if not self.slab_enabled:
            return 0
        sig = self._param_signature()
        if self.w1_slab is None:
            self.w1_slab = torch.empty(int(self.lib.gmc_w1_slab_floats(self.N, self.Fp)), dtype=torch.float32,
                                       device=self.device)
            self._slab_sig = None
        if sig != self._slab_sig:
            rc = self.lib.gmc_w1_slab_f32(hip.ptr(self.flat), self.N, self.Fp, hip.ptr(self.w1_slab), hip.stream())
            hip.check(rc, "gmc_w1_slab_f32")
            self._slab_sig = sig
        return hip.ptr(self.w1_slab)

    def _model_ref(self, slab: bool):
        """byref(gmc_model) for a call; ``slab``: with the (current) slab copy of W1."""
        self._model.W1_slab = self.ensure_slab() if slab else None
        return C.byref(self._model)

    def set_dropout(self, p: float, seed: Optional[int] = None) -> None:
        """F.dropout between the layers (TrainingNeural.py:82) for the next forward / training calls:
        ``p`` = 0 switches it off (eval mode, every reference configuration).  ``seed`` = None draws one
        from torch's CPU generator, so ``torch.manual_seed`` makes runs repeatable (the mask itself is the
        library's counter-based hash, not torch's random stream)."""
        p = float(p)
        if not 0.0 <= p < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
        if p > 0.0 and seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._model.dropout_p = p
        self._model.dropout_seed_lo = (seed or 0) & 0xFFFFFFFF
        self._model.dropout_seed_hi = ((seed or 0) >> 32) & 0xFFFFFFFF

    def dropout_state(self) -> Tuple[float, int]:
        return float(self._model.dropout_p), (int(self._model.dropout_seed_hi) << 32) | int(self._model.dropout_seed_lo)

    def adopt(self, module: torch.nn.Module) -> None:
        """Make ``module.conv{1,2}.{weight,bias}`` views of the flat buffer (values kept)."""
        named = dict(module.named_parameters())
        views = self.views()
        for k in PARAM_ORDER:
            p = named[k]
            if p.data_ptr() != views[k].data_ptr() or p.device != self.device or p.stride() != views[k].stride():
                views[k].copy_(p.detach().to(self.device, torch.float32))
                p.data = views[k]
        self._adopted = [named[k] for k in PARAM_ORDER]
        self._slab_sig = None

    def owns(self, module: torch.nn.Module) -> bool:
        named = dict(module.named_parameters())
        views = self.views()
        return all(named[k].data_ptr() == views[k].data_ptr() and named[k].stride() == views[k].stride()
                   for k in PARAM_ORDER)

    def make_batch(self, handles, values=None) -> GraphBatch:
        return GraphBatch(handles, values, self.device)

    # ---- scratch
    def _workspace(self, batch: GraphBatch, training: bool) -> Tuple[torch.Tensor, int]:
        need = int(self.lib.gmc_workspace_bytes(batch.ref(), C.byref(self._model), int(training)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        return self._ws, self._ws.numel()

    # ---- compute
    def workspace_bytes(self, batch: GraphBatch, training: bool) -> int:
        return max(256, int(self.lib.gmc_workspace_bytes(batch.ref(), C.byref(self._model), int(training))))

    def forward(self, batch: GraphBatch, C_: float = 1.0, want_loss: bool = False,
                ws: Optional[torch.Tensor] = None):
        """P [R,3] (and S [R], loss [B] when ``want_loss``) - TrainingNeural.py:79-85.
        ``ws``: caller-owned scratch (kept alive for a later :meth:`backward_from_gp`)."""
        P = torch.empty((batch.R, 3), dtype=torch.float32, device=self.device)
        S = torch.empty(batch.R, dtype=torch.int32, device=self.device) if want_loss else None
        loss = torch.empty(batch.B, dtype=torch.float32, device=self.device) if want_loss else None
        if batch.B == 0:   # nothing to launch (empty tensors have no device pointer to hand over)
            return P, S, loss
        ws, nbytes = (ws, ws.numel()) if ws is not None else self._workspace(batch, False)
        rc = self.lib.gmc_forward(batch.ref(), C.byref(self._model), C_, hip.ptr(ws), nbytes,
                                  hip.ptr(P), hip.ptr(S), hip.ptr(loss), hip.stream())
        hip.check(rc, "gmc_forward")
        return P, S, loss

    def train_fwd_bwd(self, batch: GraphBatch, C_: float = 1.0, out=None, ws: Optional[torch.Tensor] = None,
                      slab: bool = False):
        """forward + loss + backward for the batch's summed loss; gradient lands in
        ``self.grad[:count]`` - TrainingNeural.py:373-385.  ``ws``: caller-owned scratch (a trainer
        whose launches are captured into a hipGraph must own it: the engine's own scratch moves
        whenever a later call needs more)."""
        if out is None:
            P = torch.empty((batch.R, 3), dtype=torch.float32, device=self.device)
            S = torch.empty(batch.R, dtype=torch.int32, device=self.device)
            loss = torch.empty(batch.B, dtype=torch.float32, device=self.device)
        else:
            P, S, loss = out
        if batch.B == 0:   # no graphs: zero gradient AND zero loss in the tail slot, nothing to launch
            self.grad[:self.count + 1].zero_()
            return P, S, loss
        ws, nbytes = (ws, ws.numel()) if ws is not None else self._workspace(batch, True)
        rc = self.lib.gmc_train_fwd_bwd(batch.ref(), self._model_ref(slab), C_, hip.ptr(ws), nbytes,
                                        hip.ptr(P), hip.ptr(S), hip.ptr(loss), hip.ptr(self.grad),
                                        hip.stream())
        self._model.W1_slab = None
        hip.check(rc, "gmc_train_fwd_bwd")
        return P, S, loss

    def train_step(self, batch: GraphBatch, lr: float, C_: float = 1.0, out=None, betas=(0.9, 0.999),
                   eps: float = 1e-8, ws: Optional[torch.Tensor] = None, slab: bool = False,
                   loss_ptr: Optional[int] = None):
        """One whole optimizer step (forward, loss, backward, fused gradient fold + Adam) - the
        single-GPU form of the loop body of train_single_epoch (TrainingNeural.py:373-386).
        Replay-invariant: the step number is read from / advanced in device memory.  ``loss_ptr``: device-side
        address of pinned host memory (``hip.mapped_ptr``) that receives the per-graph losses instead of
        ``out[2]`` - each is stored as soon as it is final, before the backward kernels run."""
        ws, nbytes = (ws, ws.numel()) if ws is not None else self._workspace(batch, True)
        if out is None:
            P = torch.empty((batch.R, 3), dtype=torch.float32, device=self.device)
            S = torch.empty(batch.R, dtype=torch.int32, device=self.device)
            loss = torch.empty(batch.B, dtype=torch.float32, device=self.device)
        else:
            P, S, loss = out
        tail = (self.ensure_slab() if slab else None, hip.stream())
        rc = self.lib.gmc_train_step_f32(batch.ref(), self.N, self.Fp, hip.ptr(self.flat), C_, hip.ptr(ws), nbytes,
                                         hip.ptr(P), hip.ptr(S), loss_ptr or hip.ptr(loss), hip.ptr(self.grad),
                                         hip.ptr(self.m), hip.ptr(self.v), lr, betas[0], betas[1], eps,
                                         hip.ptr(self.step_dev), *tail)
        if not slab:
            self._slab_sig = None   # W1 moved, the copy did not
        hip.check(rc, "gmc_train_step_f32")
        self.step_count += 1
        self._dev_step += 1
        return P, S, loss

    def backward_from_gp(self, batch: GraphBatch, P: torch.Tensor, GP: torch.Tensor,
                         ws: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Gradients for a caller-supplied dLoss/dP (autograd path); ``ws`` must be the
        (training-sized) workspace of the forward that produced ``P``."""
        ws, nbytes = (ws, ws.numel()) if ws is not None else self._workspace(batch, True)
        rc = self.lib.gmc_backward_from_gp(batch.ref(), C.byref(self._model), hip.ptr(ws), nbytes,
                                           hip.ptr(P), hip.ptr(GP.contiguous()), hip.ptr(self.grad),
                                           hip.stream())
        hip.check(rc, "gmc_backward_from_gp")
        return self.views(self.grad)

    def adam_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        """torch.optim.Adam.step over the flat buffer (TrainingNeural.py:386)."""
        self.step_count += 1
        rc = self.lib.gmc_adam_f32(hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.m),
                                   hip.ptr(self.v), self.count, lr, betas[0], betas[1], eps,
                                   self.step_count, hip.stream())
        hip.check(rc, "gmc_adam_f32")
        self._slab_sig = None   # W1 moved, the slab copy did not

    def adam_step_dev(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, slab: bool = False,
                      publish: Optional[Tuple[torch.Tensor, int]] = None) -> None:
        """Same update with the step number read from (and advanced in) device memory, so the
        launch can be captured into a hipGraph and replayed (``step_dev`` must equal
        ``step_count`` on entry; both advance by one).  ``slab``: the updated conv1.weight goes to the slab
        copy as well (which must be current: :meth:`ensure_slab`).  ``publish`` = (device floats, device-side
        address of pinned host memory): the values are stored there by the one-wave launch that also advances the
        step counter, in front of the Adam sweep (two launches for publish + Adam + tick instead of three)."""
        if publish is not None:
            src, dst = publish
            rc = self.lib.gmc_publish_adam_devstep_model_f32(
                hip.ptr(src), src.numel(), dst, hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.m), hip.ptr(self.v),
                self.N, self.Fp, self.ensure_slab() if (slab and self.slab_enabled) else None, lr, betas[0], betas[1], eps,
                hip.ptr(self.step_dev), hip.stream())
            if not (slab and self.slab_enabled):
                self._slab_sig = None
        elif slab and self.slab_enabled:
            rc = self.lib.gmc_adam_devstep_model_f32(hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.m),
                                                     hip.ptr(self.v), self.N, self.Fp, self.ensure_slab(), lr, betas[0],
                                                     betas[1], eps, hip.ptr(self.step_dev), hip.stream())
        else:
            rc = self.lib.gmc_adam_devstep_f32(hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.m),
                                               hip.ptr(self.v), self.count, lr, betas[0], betas[1], eps,
                                               hip.ptr(self.step_dev), hip.stream())
            self._slab_sig = None
        hip.check(rc, "gmc_adam_devstep_f32")
        self.step_count += 1
        self._dev_step += 1

    def sync_step_dev(self) -> None:
        """Make the device-side step counter equal ``step_count`` - a fill launch only when they differ
        (after host-stepped Adam launches); a trainer replaying its graph epoch after epoch never needs it."""
        if self._dev_step != self.step_count:
            self.step_dev.fill_(self.step_count)
            self._dev_step = self.step_count

    def sync_replicas(self, src: int = 0) -> None:
        """Data-parallel start-up: every rank takes rank ``src``'s parameters, Adam moments and step
        count, so that replicas built from different RNG states apply the all-reduced gradient to the
        SAME weights (the reference has one model; N replicas must be that one model)."""
        if not dp_active():
            return
        for buf in (self.flat, self.m, self.v):
            dist.broadcast(buf, src)
        step = torch.tensor([self.step_count], dtype=torch.int64, device=self.device)
        dist.broadcast(step, src)
        self.step_count = int(step.item())
        self._dev_step = -1
        self._slab_sig = None   # (a collective writes through raw pointers: no version counter moves)
        self.sync_step_dev()

    def publish(self, src: torch.Tensor, pinned_dst: int) -> None:
        """``src`` (device floats) -> pinned host memory at its device-side address ``pinned_dst``
        (``hip.mapped_ptr``), one system-scope store per value, on the current stream."""
        rc = self.lib.gmc_publish_f32(hip.ptr(src), src.numel(), pinned_dst, hip.stream())
        hip.check(rc, "gmc_publish_f32")

    def allreduce_grad(self, local_loss_sum: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """One RCCL all-reduce (sum) of [grad | loss] over xGMI when torch.distributed is up."""
        if local_loss_sum is not None:
            self.grad[self.count] = local_loss_sum
        if dp_active():
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        return self.grad[self.count] if local_loss_sum is not None else None


def dp_active() -> bool:
    """Do steps run the data-parallel sequence (shard step -> all-reduce -> Adam)?  Yes with more than one
    rank; also with ONE rank when GCN_MAXCUT_DP_SINGLE_RANK=1 - a one-GPU box then exercises RCCL itself
    (communicator set-up, the collective on the launch stream, hipGraph capture beside its watchdog)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GCN_MAXCUT_DP_SINGLE_RANK") == "1"


def shard_for_rank(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced slice of the dataset for one rank."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_by_weight(weights: Sequence[int], rank: int, world: int) -> range:
    """Contiguous slice of a group of graphs for one rank, balanced by WORK instead of by count (SURVEY section 8e:
    "for mixed sizes balance by nnz"): ``weights[i]`` = directed edges of graph i.  Equal weights (the regular-graph
    workloads) give exactly :func:`shard_for_rank`.  Otherwise the partition into ``world`` contiguous runs with the
    smallest possible largest run: the smallest capacity for which filling rank after rank (a rank closes when the
    next graph would exceed it) needs at most ``world`` ranks, then that filling - a pure function of the weights, so
    every rank computes the same split without talking; trailing ranks may stay empty."""
    w = [int(x) for x in weights]
    n = len(w)
    if n == 0 or world <= 1 or min(w) == max(w):
        return shard_for_rank(n, rank, world)

    def cuts(cap):
        out, load = [0], 0
        for i, x in enumerate(w):
            if load and load + x > cap:
                out.append(i)
                load = 0
            load += x
        return out

    lo, hi = max(w), sum(w)
    while lo < hi:
        mid = (lo + hi) // 2
        if len(cuts(mid)) <= world:
            hi = mid
        else:
            lo = mid + 1
    starts = cuts(lo)
    starts += [n] * (world + 1 - len(starts))
    return range(starts[rank], starts[rank + 1])
