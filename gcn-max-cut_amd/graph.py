"""Graph handle and HBM-resident batches (host side of ``gmc_batch``).

``GraphHandle`` stands where the reference stores a ``DGLGraph``
(``python/DataGenerator/graphExtender.py:102-103,114``): entry 0 of every dataset
item.  It answers what the reference and its notebooks ask of that object
(``number_of_nodes``, ``number_of_edges`` == 2|E|, ``.to(device)``, picklable) and owns
the CSR the HIP kernels aggregate over.

``GraphBatch`` is a block-diagonal batch of graphs laid out for the kernels: one
``rowptr``/``gcol``/``lcol``/``vals``/``dinv`` set in HBM plus per-graph row offsets.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import hip


class DGLError(Exception):
    """Name-compatible stand-in for the error DGL raises (zero in-degree nodes)."""


class GraphHandle:
    """CSR of an undirected graph as ``dgl.from_networkx`` would hold it: node ids are
    the sorted networkx labels, every undirected edge appears in both directions."""

    def __init__(self, n: int, rowptr: np.ndarray, col: np.ndarray, weight: Optional[np.ndarray] = None):
        self.n = int(n)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        # edge 'weight' attribute in CSR order (what the padded adjacency holds), or None = all 1
        self.weight = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
        self.device = torch.device("cpu")
        self._cache = {}

    # ---- the DGLGraph surface the reference touches
    def number_of_nodes(self) -> int:
        return self.n

    def number_of_edges(self) -> int:
        return int(self.col.size)

    num_nodes = number_of_nodes
    num_edges = number_of_edges

    def in_degrees(self) -> torch.Tensor:
        return torch.from_numpy(np.diff(self.rowptr).astype(np.int64))

    out_degrees = in_degrees

    def to(self, device) -> "GraphHandle":
        self.device = torch.device(device)
        return self

    def edges(self):
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.rowptr))
        return torch.from_numpy(self.col.astype(np.int64)), torch.from_numpy(rows)  # (src, dst)

    def __repr__(self) -> str:
        return f"GraphHandle(num_nodes={self.n}, num_edges={self.col.size})"

    def __getstate__(self):
        return {"n": self.n, "rowptr": self.rowptr, "col": self.col, "weight": self.weight}

    def __setstate__(self, st):
        self.__init__(st["n"], st["rowptr"], st["col"], st.get("weight"))

    # ---- kernels' view
    def row_index(self) -> np.ndarray:
        r = self._cache.get("rows")
        if r is None:
            r = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.rowptr))
            self._cache["rows"] = r
        return r

    def check_degrees(self) -> None:
        if self.n and int(np.diff(self.rowptr).min()) == 0:
            raise DGLError(
                "There are 0-in-degree nodes in the graph, output for those nodes will be invalid "
                "(GraphConv, allow_zero_in_degree=False)")

    def edge_values(self, inputs: Optional[torch.Tensor]) -> Optional[np.ndarray]:
        """Values the feature matrix ``inputs`` ([n, N], the padded adjacency) holds on the
        graph's edges, in CSR order; ``None`` when they are all exactly 1.  ``inputs`` must be
        zero off the edges (true for every call site of the reference:
        TrainingNeural.py:373,555; TestingNeuralNetwork.py:143)."""
        if inputs is None:
            w = self.weight
        else:
            key = ("vals", inputs.data_ptr(), inputs._version, tuple(inputs.shape), str(inputs.device))
            if key in self._cache:
                return self._cache[key]
            if inputs.dim() != 2 or inputs.shape[0] != self.n:
                raise ValueError(f"features must be [n={self.n}, N], got {tuple(inputs.shape)}")
            if inputs.shape[1] < self.n:
                raise ValueError("N should be greater than or equal to the original matrix size.")
            rows = torch.from_numpy(self.row_index()).to(inputs.device)
            cols = torch.from_numpy(self.col.astype(np.int64)).to(inputs.device)
            vals = inputs[rows, cols]
            if int(torch.count_nonzero(inputs)) != int(torch.count_nonzero(vals)):
                raise NotImplementedError(
                    "features with non-zeros off the graph's edges: this path implements the "
                    "reference's usage net(g, padded_adjacency) (TrainingNeural.py:373)")
            w = vals.detach().to("cpu", torch.float32).numpy()
            self._cache.clear()
            self._cache[key] = None if bool(np.all(w == 1.0)) else w
            return self._cache[key]
        if w is None or bool(np.all(w == 1.0)):
            return None
        return w


def from_networkx(nx_graph) -> GraphHandle:
    """``dgl.from_networkx`` for undirected graphs (graphExtender.py:102): labels sorted
    to 0..n-1, both directions of each edge, plus the ``weight`` attribute per edge."""
    if nx_graph.is_directed():
        raise NotImplementedError("directed graphs are outside the reference's data path")
    nodes = sorted(nx_graph.nodes())
    n = len(nodes)
    identity = nodes == list(range(n))
    index = None if identity else {u: i for i, u in enumerate(nodes)}
    m = nx_graph.number_of_edges()
    src = np.empty(m, np.int64); dst = np.empty(m, np.int64); w = np.empty(m, np.float32)
    for i, (u, v, wt) in enumerate(nx_graph.edges(data="weight", default=1)):
        src[i] = u if identity else index[u]
        dst[i] = v if identity else index[v]
        w[i] = wt
    loops = src == dst
    rows = np.concatenate([src, dst[~loops]])
    cols = np.concatenate([dst, src[~loops]])
    ww = np.concatenate([w, w[~loops]])
    order = np.lexsort((cols, rows))
    rows, cols, ww = rows[order], cols[order], ww[order]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return GraphHandle(n, rowptr.astype(np.int32), cols.astype(np.int32),
                       None if bool(np.all(ww == 1.0)) else ww)


class BatchArrays:
    """Host-side (numpy) form of a block-diagonal batch: everything ``struct gmc_batch``
    points to, built without touching a GPU (so the layout is testable on CPU)."""

    def __init__(self, handles: Sequence[GraphHandle], values: Optional[Sequence[Optional[np.ndarray]]] = None):
        B = len(handles)
        ns = np.asarray([h.n for h in handles], np.int64)
        for h in handles:
            if h.n < 3 or h.n > hip.MAX_GRAPH_NODES:
                raise ValueError(f"graph with {h.n} nodes: need 3 <= n <= {hip.MAX_GRAPH_NODES}")
            h.check_degrees()
        goff = np.zeros(B + 1, np.int64)
        np.cumsum(ns, out=goff[1:])
        nnzs = np.asarray([h.col.size for h in handles], np.int64)
        eoff = np.zeros(B + 1, np.int64)
        np.cumsum(nnzs, out=eoff[1:])
        if goff[-1] >= 2 ** 31 or eoff[-1] >= 2 ** 31:
            raise ValueError("batch too large for int32 indices")
        rowptr = np.zeros(int(goff[-1]) + 1, np.int32)
        lcol = np.empty(int(eoff[-1]), np.int32)
        gcol = np.empty(int(eoff[-1]), np.int32)
        if values is None:
            values = [h.edge_values(None) for h in handles]
        unit = all(v is None for v in values)
        vals = None if unit else np.ones(int(eoff[-1]), np.float32)
        for g, h in enumerate(handles):
            rowptr[goff[g] + 1: goff[g + 1] + 1] = h.rowptr[1:] + eoff[g]
            lcol[eoff[g]:eoff[g + 1]] = h.col
            gcol[eoff[g]:eoff[g + 1]] = h.col + goff[g]
            if vals is not None and values[g] is not None:
                vals[eoff[g]:eoff[g + 1]] = values[g]
        degi = np.diff(rowptr)
        dinv = (1.0 / np.sqrt(np.maximum(degi.astype(np.float32), 1.0))).astype(np.float32)
        # ELL copy for the LDS-tiled kernels: W slots per row (8 or 16) holding the row's first W neighbours
        # (CSR order, then re-arranged over the slots by the library), padded with the graph's node count (the id
        # of an all-zero tile row) / weight 0.  Neighbours beyond the W-th of a row go to the overflow lists
        # (blocks of eight ids per row): a hub node costs its own extra blocks, the batch stays on the LDS path.
        ell = ell_vals = None
        ovf_ptr = ovf_ids = ovf_vals = None
        ovf_max_blocks = 0
        ell_slots = 0
        max_deg = int(degi.max()) if degi.size else 0
        R = int(goff[-1])
        W = self.choose_width(degi)
        if W and int(ns.max()) < 65535:
            deg64 = degi.astype(np.int64)
            slot = np.arange(int(eoff[-1]), dtype=np.int64) - np.repeat(rowptr[:-1].astype(np.int64), deg64)
            rows = np.repeat(np.arange(R, dtype=np.int64), deg64)
            n_of_row = np.repeat(ns, ns)
            inside = slot < W
            ell = n_of_row.astype(np.uint16)[:, None].repeat(W, axis=1)
            ell[rows[inside], slot[inside]] = lcol[inside].astype(np.uint16)
            if vals is not None:
                ell_vals = np.zeros((R, W), np.float32)
                ell_vals[rows[inside], slot[inside]] = vals[inside]
            if max_deg > W:
                blocks = (np.maximum(deg64 - W, 0) + 7) // 8
                ovf_ptr = np.zeros(R + 1, np.int64)
                np.cumsum(blocks, out=ovf_ptr[1:])
                # what the kernels' 16-bit row descriptors can say: <= 15 blocks per row (degree <= W + 120), <= 4095
                # blocks per graph; beyond that the batch takes the row kernels
                ovf_max_blocks = int(np.diff(ovf_ptr[goff]).max())
                if int(blocks.max()) > 15 or ovf_max_blocks > 4095:
                    W = 0
            if W and max_deg > W:
                nblk = int(ovf_ptr[-1])
                ovf_ids = np.repeat(n_of_row.astype(np.uint16), blocks * 8)       # padding: the zero row of the graph
                where = ovf_ptr[:-1][rows[~inside]] * 8 + (slot[~inside] - W)
                ovf_ids[where] = lcol[~inside].astype(np.uint16)
                if vals is not None:
                    ovf_vals = np.zeros(nblk * 8, np.float32)
                    ovf_vals[where] = vals[~inside]
                ovf_ptr = ovf_ptr.astype(np.int32)
            if not W:
                ell = ell_vals = ovf_ptr = None
                ovf_max_blocks = 0
        if W and int(ns.max()) < 65535:
            # LDS-bank-aware slot order (host routine of the library; plain CSR order otherwise)
            try:
                lib = hip.load()
            except hip.HipExtensionError:
                lib = None
            if lib is not None:
                ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
                go32 = goff.astype(np.int32)
                rc = lib.gmc_ell_arrange_host(B, ptr(go32), ptr(rowptr), ptr(lcol), ptr(vals), W, ptr(ell), ptr(ell_vals))
                hip.check(rc, "gmc_ell_arrange_host")
                # s < W when no row of the batch has more than s neighbours: slots s.. of every row are then padding
                ell_slots = int(lib.gmc_ell_slots_for(R, ptr(rowptr), W))
        self.B, self.R, self.nnz = B, int(goff[-1]), int(eoff[-1])
        self.n_max = int(ns.max()) if B else 0
        self.nnz_max = int(nnzs.max()) if B else 0
        self.uniform_n = int(ns[0]) if B and bool(np.all(ns == ns[0])) else 0
        self.sizes, self.goff = ns, goff
        self.rowptr, self.gcol, self.lcol, self.vals, self.dinv = rowptr, gcol, lcol, vals, dinv
        self.ell, self.ell_vals, self.ell_width = ell, ell_vals, (W if ell is not None else 0)
        self.ell_slots = ell_slots if ell is not None else 0   # (without the library: CSR slot order, every slot live)
        self.ovf_ptr, self.ovf_ids, self.ovf_vals = ovf_ptr, ovf_ids, ovf_vals
        self.ovf_max_blocks = ovf_max_blocks if ovf_ptr is not None else 0
        self.max_degree = max_deg

    @staticmethod
    def choose_width(degi: np.ndarray) -> int:
        """Slots per row of the ELL table for rows of these degrees; 0 = no table (row kernels).  8 when (almost)
        every row has at most 8 neighbours - up to one row in 200 may be longer, its extra neighbours become
        overflow blocks - else 16 with the same allowance; no table once the overflow lists would hold more than
        an eighth of the edges (dense graphs: the row kernels read those from HBM / L2 anyway)."""
        if degi.size == 0 or int(degi.max()) <= 0:
            return 0
        R, nnz = degi.size, int(degi.sum())
        for W in (8, 16):
            over = degi > W
            if int(over.sum()) * 200 <= R or (W == 16 and int((degi[over] - W).sum()) * 8 <= nnz):
                return W
        return 0


class GraphBatch:
    """Block-diagonal batch resident on one GPU; mirrors ``struct gmc_batch``."""

    def __init__(self, handles: Sequence[GraphHandle], values: Optional[Sequence[Optional[np.ndarray]]] = None,
                 device: Optional[torch.device] = None):
        device = device or hip.require_gpu()
        self.device = device
        h = BatchArrays(handles, values)
        self.host = h
        self.B, self.R, self.nnz, self.n_max, self.uniform_n = h.B, h.R, h.nnz, h.n_max, h.uniform_n
        self.sizes, self.goff_host = h.sizes, h.goff
        dev = lambda a: None if a is None else torch.from_numpy(a).to(device)
        self.goff = dev(h.goff.astype(np.int32))
        self.rowptr, self.gcol, self.lcol = dev(h.rowptr), dev(h.gcol), dev(h.lcol)
        self.vals, self.dinv = dev(h.vals), dev(h.dinv)
        self.ell = None if h.ell is None else torch.from_numpy(h.ell.view(np.int16)).to(device)
        self.ell_vals = dev(h.ell_vals)
        self.ovf_ptr = dev(h.ovf_ptr)
        self.ovf_ids = None if h.ovf_ids is None else torch.from_numpy(h.ovf_ids.view(np.int16)).to(device)
        self.ovf_vals = dev(h.ovf_vals)
        self.c = hip.GmcBatch(
            B=h.B, R=h.R, nnz=h.nnz, n_max=h.n_max, uniform_n=h.uniform_n, nnz_max=h.nnz_max,
            goff=hip.ptr(self.goff), rowptr=hip.ptr(self.rowptr), gcol=hip.ptr(self.gcol),
            lcol=hip.ptr(self.lcol), vals=hip.ptr(self.vals), dinv=hip.ptr(self.dinv),
            ell=hip.ptr(self.ell), ell_vals=hip.ptr(self.ell_vals), ell_width=h.ell_width, ell_slots=h.ell_slots,
            ovf_ptr=hip.ptr(self.ovf_ptr), ovf_ids=hip.ptr(self.ovf_ids), ovf_vals=hip.ptr(self.ovf_vals),
            ovf_max_blocks=h.ovf_max_blocks)

    def ref(self):
        return C.byref(self.c)

    def split(self, t: torch.Tensor) -> List[torch.Tensor]:
        """Per-graph views of a [R, ...] tensor."""
        return [t[int(self.goff_host[g]):int(self.goff_host[g + 1])] for g in range(self.B)]

    def spmm_bytes(self, F: int) -> int:
        """Algorithmic HBM bytes of one F-wide SpMM over this batch."""
        return 2 * self.R * F * 4 + self.nnz * 4 + (self.R + 1) * 4 + 2 * self.R * 4 + F * 4
