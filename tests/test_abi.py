"""CPU-side checks of the C-ABI boundary: the in-tree library loads without a GPU and
exports every symbol include/gcnmaxcut.h declares; the product path refuses to compute
without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch


def header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "gcnmaxcut.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = built.hip.load()
    names = header_symbols()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(built.hip.SYMBOLS) == names
    assert lib.gmc_version() == 100
    assert lib.gmc_error_string(-3).decode().startswith("number_classes must be 3")


def test_header_cites_reference_lines():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "gcnmaxcut.h")).read()
    assert text.count("TrainingNeural.py") >= 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback(built):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    from tests import util
    ds = util.product_dataset([(30, 5, 1)])
    net, _, _ = T.setup_model_and_optimizer(T.TrainingConfig(n_nodes=1000, hidden_dim=16))
    (g, a_pad, _nx, _t), = ds.values()
    with pytest.raises(built.hip.HipExtensionError):
        net(g, a_pad)
    with pytest.raises(built.hip.HipExtensionError):
        T.evaluate_model(net, ds, T.TrainingConfig())
    with pytest.raises(built.hip.HipExtensionError):
        built.GraphBatch([g], None)


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "gcn-max-cut_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "never", (dirpath, f)


def test_argument_errors_need_no_gpu(built):
    """The entry points added for the slab copy of W1 / host-memory losses reject bad arguments with their status
    codes before any HIP call (so this runs without a GPU); sizes come from a host function."""
    import ctypes as C
    lib = built.hip.load()
    assert lib.gmc_w1_slab_floats(1000, 500) == 32 * 1000 * 16        # [ceil(F/16)][N][16]
    assert lib.gmc_w1_slab_floats(1000, 72) == 5 * 1000 * 16
    assert lib.gmc_w1_slab_floats(0, 500) == 0
    null, some = C.c_void_p(None), C.c_void_p(4096)                    # (never dereferenced: the calls fail first)
    assert lib.gmc_w1_slab_f32(null, 1000, 500, some, None) == -1      # GMC_ERR_NULL
    assert lib.gmc_w1_slab_f32(some, 1000, 501, some, None) == -2      # GMC_ERR_SHAPE: F % 4
    assert lib.gmc_w1_slab_f32(some, 1000, 500, C.c_void_p(4100), None) == -4   # GMC_ERR_ALIGN
    assert lib.gmc_publish_f32(null, 4, some, None) == -1
    assert lib.gmc_publish_f32(some, -1, some, None) == -2
    assert lib.gmc_publish_f32(some, 0, some, None) == 0               # nothing to do, nothing launched
    assert lib.gmc_host_device_pointer(null, C.byref(C.c_void_p())) == -1
    assert lib.gmc_adam_devstep_model_f32(some, some, some, some, 1000, 501, null, 1e-3, 0.9, 0.999, 1e-8, some, None) == -2
    assert lib.gmc_adam_devstep_model_f32(some, some, some, some, 1000, 500, C.c_void_p(4100), 1e-3, 0.9, 0.999, 1e-8,
                                          some, None) == -4
