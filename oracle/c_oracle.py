"""TEST INFRASTRUCTURE ONLY - ctypes front-end of ``oracle/gcn_oracle.c``.

Loaded by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg; the
product path never imports it.  See the C file's header for the reference lines each
function follows and for the parity status (unpinned at the DGL boundary).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgcn_oracle.so")
_lib = None

_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gcn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def csr_of(nx_graph):
    """CSR (sorted columns) of the symmetric adjacency with ``weight`` values."""
    n = nx_graph.number_of_nodes()
    nbrs = [[] for _ in range(n)]
    for u, v, w in nx_graph.edges(data="weight", default=1):
        nbrs[u].append((v, float(w)))
        nbrs[v].append((u, float(w)))
    rowptr = np.zeros(n + 1, np.int32)
    col, val = [], []
    for r in range(n):
        nbrs[r].sort()
        rowptr[r + 1] = rowptr[r] + len(nbrs[r])
        col += [c for c, _ in nbrs[r]]
        val += [w for _, w in nbrs[r]]
    return rowptr, np.asarray(col, np.int32), np.asarray(val, np.float32)


def spmm(rowptr, col, val, scale, X, bias, relu):
    n, F = rowptr.size - 1, X.shape[1]
    Y = np.empty((n, F), np.float32)
    X = np.ascontiguousarray(X, np.float32)
    lib().orc_spmm(C.c_int(n), _opt(rowptr), _opt(col), _opt(val), _opt(scale), _opt(X),
                   C.c_long(X.shape[1]), _opt(bias), C.c_int(int(relu)), _opt(Y), C.c_long(F),
                   C.c_int(F))
    return Y


def forward(rowptr, col, val, W1, b1, W2, b2):
    n, F, K = rowptr.size - 1, W1.shape[1], W2.shape[1]
    T0 = np.empty((n, F), np.float32); H = np.empty((n, F), np.float32)
    Z0 = np.empty((n, K), np.float32); P = np.empty((n, K), np.float32)
    rc = lib().orc_forward(C.c_int(n), _opt(rowptr), _opt(col), _opt(val), C.c_int(F), C.c_int(K),
                           _opt(W1), _opt(b1), _opt(W2), _opt(b2), _opt(T0), _opt(H), _opt(Z0),
                           _opt(P))
    if rc:
        raise ValueError(f"orc_forward rc={rc}")
    return dict(T0=T0, H=H, Z0=Z0, P=P)


def loss_grad(rowptr, col, val, P, Cc=1.0):
    n, K = P.shape
    S = np.empty(n, np.int32); GP = np.empty((n, K), np.float32); loss = np.zeros(1, np.float32)
    lib().orc_loss_grad(C.c_int(n), _opt(rowptr), _opt(col), _opt(val), C.c_int(K), _opt(P),
                        C.c_float(Cc), _opt(S), _opt(loss), _opt(GP))
    return S, float(loss[0]), GP


def backward(rowptr, col, val, N, W2, H, P, GP):
    n, F = H.shape
    K = P.shape[1]
    dW1 = np.zeros((N, F), np.float32); db1 = np.zeros(F, np.float32)
    dW2 = np.zeros((F, K), np.float32); db2 = np.zeros(K, np.float32)
    rc = lib().orc_backward(C.c_int(n), _opt(rowptr), _opt(col), _opt(val), C.c_int(F), C.c_int(K),
                            _opt(W2), _opt(H), _opt(P), _opt(GP), _opt(dW1), _opt(db1), _opt(dW2),
                            _opt(db2))
    if rc:
        raise RuntimeError(f"orc_backward rc={rc}")
    return dW1, db1, dW2, db2


def adam(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    lib().orc_adam(_opt(p), _opt(g), _opt(m), _opt(v), C.c_long(p.size), C.c_double(lr),
                   C.c_double(beta1), C.c_double(beta2), C.c_double(eps), C.c_int(step))


class CTrainer:
    """Flat-parameter trainer over ``orc_train_step`` ([W1 | b1 | W2 | b2])."""

    def __init__(self, params: Dict[str, np.ndarray], lr=1e-3, Cc=1.0):
        W1, W2 = params["conv1.weight"], params["conv2.weight"]
        self.N, self.F = W1.shape
        self.K = W2.shape[1]
        self.flat = np.concatenate([np.asarray(params[k], np.float32).ravel() for k in
                                    ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias")])
        self.grad = np.zeros_like(self.flat)
        self.m = np.zeros_like(self.flat)
        self.v = np.zeros_like(self.flat)
        self.lr, self.C, self.t = lr, Cc, 0

    def step(self, csrs: Sequence) -> np.ndarray:
        """``csrs`` = [(rowptr, col, val_or_None), ...]; returns per-graph losses."""
        B = len(csrs)
        n_of = np.asarray([c[0].size - 1 for c in csrs], np.int32)
        PP = C.c_void_p * B
        rp = PP(*[c[0].ctypes.data for c in csrs])
        cl = PP(*[c[1].ctypes.data for c in csrs])
        unit = all(c[2] is None for c in csrs)
        vl = None if unit else PP(*[c[2].ctypes.data for c in csrs])
        losses = np.zeros(B, np.float32)
        self.t += 1
        rc = lib().orc_train_step(C.c_int(B), _opt(n_of), rp, cl, vl, C.c_int(self.N),
                                  C.c_int(self.F), C.c_int(self.K), _opt(self.flat),
                                  _opt(self.grad), _opt(self.m), _opt(self.v), C.c_double(self.lr),
                                  C.c_float(self.C), C.c_int(self.t), _opt(losses))
        if rc:
            raise RuntimeError(f"orc_train_step rc={rc}")
        return losses

    def unpack(self) -> Dict[str, np.ndarray]:
        N, F, K = self.N, self.F, self.K
        o = np.cumsum([0, N * F, F, F * K, K])
        f = self.flat
        return {"conv1.weight": f[o[0]:o[1]].reshape(N, F).copy(), "conv1.bias": f[o[1]:o[2]].copy(),
                "conv2.weight": f[o[2]:o[3]].reshape(F, K).copy(), "conv2.bias": f[o[3]:o[4]].copy()}
