"""ctypes binding of ``libgcnmaxcut_hip.so`` (declared in ``include/gcnmaxcut.h``).

This is the only door from the Python host code to the compute path.  There is no
CPU fallback: if the shared library is missing, or no HIP device is visible, every
compute entry point raises (see :func:`require_gpu`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCN_MAXCUT_LIB") or os.path.join(_PKG, "lib", "libgcnmaxcut_hip.so")

# every symbol include/gcnmaxcut.h declares (tests check the .so exports all of them)
SYMBOLS = (
    "gmc_version", "gmc_error_string", "gmc_spmm_f32", "gmc_dense_hw2_f32", "gmc_head_f32",
    "gmc_adam_f32", "gmc_workspace_bytes", "gmc_forward", "gmc_train_fwd_bwd",
    "gmc_backward_from_gp", "gmc_probe_begin", "gmc_probe_end", "gmc_set_fuse", "gmc_decode_sample_f32", "gmc_adam_devstep_f32", "gmc_ell_arrange_host", "gmc_ell_slots_for", "gmc_train_step_f32",
    "gmc_adam_devstep_model_f32", "gmc_w1_slab_floats", "gmc_w1_slab_f32", "gmc_host_device_pointer", "gmc_publish_f32",
    "gmc_publish_adam_devstep_model_f32",
)

MAX_GRAPH_NODES = 4096
MODEL_GRAD_TAIL = 1   # gmc_model.flags: grad has a tail slot that receives the batch's loss sum
ABI_VERSION = 200     # GMC_VERSION of include/gcnmaxcut.h these struct layouts follow (checked at load and per call)


class HipExtensionError(RuntimeError):
    """The HIP extension (or a GPU to run it on) is not available."""


class GmcBatch(C.Structure):
    _fields_ = [
        ("abi", C.c_int32), ("B", C.c_int32), ("R", C.c_int32), ("nnz", C.c_int32), ("n_max", C.c_int32),
        ("uniform_n", C.c_int32), ("nnz_max", C.c_int32),
        ("goff", C.c_void_p), ("rowptr", C.c_void_p), ("gcol", C.c_void_p), ("lcol", C.c_void_p),
        ("vals", C.c_void_p), ("dinv", C.c_void_p),
        ("ell", C.c_void_p), ("ell_vals", C.c_void_p), ("ell_width", C.c_int32), ("ell_slots", C.c_int32),
        ("ovf_ptr", C.c_void_p), ("ovf_ids", C.c_void_p), ("ovf_vals", C.c_void_p), ("ovf_max_blocks", C.c_int32),
    ]

    def __init__(self, **kw):
        kw.setdefault("abi", ABI_VERSION)
        super().__init__(**kw)


class GmcModel(C.Structure):
    _fields_ = [
        ("abi", C.c_int32), ("N", C.c_int32), ("F", C.c_int32), ("K", C.c_int32), ("flags", C.c_int32),
        ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
        ("dropout_p", C.c_float), ("dropout_seed_lo", C.c_uint32), ("dropout_seed_hi", C.c_uint32),
        ("W1_slab", C.c_void_p),
    ]

    def __init__(self, **kw):
        kw.setdefault("abi", ABI_VERSION)
        super().__init__(**kw)


_lib: Optional[C.CDLL] = None
HAS_SLAB = True   # (every 0.2.x library has the W1 slab entry points; kept for callers that still ask)


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    lib.gmc_version.restype = C.c_int
    lib.gmc_error_string.restype = C.c_char_p
    lib.gmc_error_string.argtypes = [C.c_int]
    lib.gmc_spmm_f32.argtypes = [vp, vp, vp, vp, vp, i64, vp, C.c_int, vp, i64, i32, i32, i32, vp, vp, vp]
    lib.gmc_dense_hw2_f32.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp]
    lib.gmc_head_f32.argtypes = [C.POINTER(GmcBatch), vp, i32, vp, f32, vp, vp, vp, vp, vp, vp]
    lib.gmc_adam_f32.argtypes = [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, i32, vp]
    lib.gmc_workspace_bytes.restype = sz
    lib.gmc_workspace_bytes.argtypes = [C.POINTER(GmcBatch), C.POINTER(GmcModel), C.c_int]
    lib.gmc_forward.argtypes = [C.POINTER(GmcBatch), C.POINTER(GmcModel), f32, vp, sz, vp, vp, vp, vp]
    lib.gmc_train_fwd_bwd.argtypes = [C.POINTER(GmcBatch), C.POINTER(GmcModel), f32, vp, sz, vp, vp, vp, vp, vp]
    lib.gmc_backward_from_gp.argtypes = [C.POINTER(GmcBatch), C.POINTER(GmcModel), vp, sz, vp, vp, vp, vp]
    lib.gmc_adam_devstep_f32.argtypes = [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp]
    lib.gmc_ell_arrange_host.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp]
    lib.gmc_ell_slots_for.argtypes = [i32, vp, i32]
    lib.gmc_train_step_f32.argtypes = [C.POINTER(GmcBatch), i32, i32, vp, f32, vp, sz, vp, vp, vp, vp, vp, vp,
                                       C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    lib.gmc_adam_devstep_model_f32.argtypes = [vp, vp, vp, vp, i32, i32, vp, C.c_double, C.c_double, C.c_double,
                                               C.c_double, vp, vp]
    lib.gmc_w1_slab_floats.argtypes = [i32, i32]
    lib.gmc_w1_slab_floats.restype = sz
    lib.gmc_w1_slab_f32.argtypes = [vp, i32, i32, vp, vp]
    lib.gmc_set_fuse.argtypes = [C.c_int]
    lib.gmc_decode_sample_f32.argtypes = [C.POINTER(GmcBatch), vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    lib.gmc_probe_begin.argtypes = [i32]
    lib.gmc_probe_end.argtypes = [vp, vp, i32]
    lib.gmc_host_device_pointer.argtypes = [vp, C.POINTER(vp)]
    lib.gmc_publish_f32.argtypes = [vp, i32, vp, vp]
    lib.gmc_publish_adam_devstep_model_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, C.c_double, C.c_double,
                                                       C.c_double, C.c_double, vp, vp]
    lib.gmc_debug_set_device_cus.argtypes = [C.c_int]   # test hook (not part of gcnmaxcut.h)
    lib.gmc_debug_set_device_cus.restype = C.c_int
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("gmc_version", "gmc_error_string", "gmc_workspace_bytes", "gmc_w1_slab_floats"):
            fn.restype = C.c_int


def load() -> C.CDLL:
    """dlopen the in-tree library (no compute; works without a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.gmc_version.restype = C.c_int
        have = int(lib.gmc_version())
        if have != ABI_VERSION:   # struct layouts and signatures are chosen by VERSION, never by sniffing symbols
            raise HipExtensionError(
                f"{LIB_PATH} reports gmc_version() = {have}, this binding is written for {ABI_VERSION} "
                f"(include/gcnmaxcut.h): rebuild the library (`python -c 'import __graft_entry__ as g; g.build()'`)")
        _declare(lib)
        _lib = lib
    return _lib


def require_gpu() -> torch.device:
    """The compute path needs the HIP library *and* a visible GPU; fail loudly otherwise."""
    load()
    if not torch.cuda.is_available():
        raise HipExtensionError(
            "no HIP device visible: the GCN max-cut compute path runs only on a GPU "
            "(MI355X/gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = load().gmc_error_string(rc).decode()
    if rc < 0:
        raise ValueError(f"{what}: {msg} (gmc error {rc})")
    raise RuntimeError(f"{what}: HIP error {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipExtensionError("tensor is not on a HIP device")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def mapped_ptr(t: torch.Tensor) -> Optional[int]:
    """Device-side address of a PINNED host tensor (kernels may store into it; the host polls it), or None when
    the library / runtime cannot map it."""
    lib = load()
    if not (t.device.type == "cpu" and t.is_pinned() and t.is_contiguous()):
        return None
    out = C.c_void_p()
    if lib.gmc_host_device_pointer(C.c_void_p(t.data_ptr()), C.byref(out)) != 0 or not out.value:
        return None
    return int(out.value)


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


KERNEL_TAGS = ("gather_w1", "agg_fwd", "head", "hidden_bwd", "colsum", "agg_bwd", "dw1", "dw1_fold",
               "adam", "spmm_user", "dense_mfma", "bwd1_fused", "fwd1_fused", "decode", "finish")


class Probe:
    """``with Probe(capacity) as p: ...`` then ``p.records`` = [(kernel_tag, ms), ...]:
    per-launch HIP-event timings recorded by the library on the launch stream."""

    def __init__(self, capacity: int):
        self.capacity, self.records = capacity, []

    def __enter__(self):
        check(load().gmc_probe_begin(self.capacity), "gmc_probe_begin")
        return self

    def __exit__(self, *exc):
        tags = (C.c_int32 * self.capacity)()
        ms = (C.c_float * self.capacity)()
        n = load().gmc_probe_end(tags, ms, self.capacity)
        if n < 0:
            raise RuntimeError(f"gmc_probe_end failed ({n})")
        self.records = [(KERNEL_TAGS[tags[i]], float(ms[i])) for i in range(min(n, self.capacity))]
        return False
