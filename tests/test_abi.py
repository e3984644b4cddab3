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
    assert lib.gmc_version() == 200 == built.hip.ABI_VERSION     # 0.2.0: structs start with an `abi` word
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gcnmaxcut.h")).read()
    assert re.search(r"#define\s+GMC_VERSION\s+200\b", header)
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


def test_structs_carry_the_abi_word_and_it_is_checked(built):
    """gmc_batch / gmc_model start with the GMC_VERSION the caller was compiled against; an entry point given a
    struct of another version refuses (GMC_ERR_ABI) before it reads any other field - no GPU needed."""
    import ctypes as C
    hip = built.hip
    lib = hip.load()
    assert hip.GmcBatch._fields_[0][0] == "abi" and hip.GmcModel._fields_[0][0] == "abi"
    assert hip.GmcBatch().abi == hip.ABI_VERSION and hip.GmcModel().abi == hip.ABI_VERSION
    names = [f[0] for f in hip.GmcBatch._fields_]
    assert names[-4:] == ["ovf_ptr", "ovf_ids", "ovf_vals", "ovf_max_blocks"] and "ell_slots" in names
    old_b, old_m = hip.GmcBatch(abi=100), hip.GmcModel(abi=100)
    some = C.c_void_p(4096)
    assert lib.gmc_workspace_bytes(C.byref(hip.GmcBatch()), C.byref(hip.GmcModel()), 1) >= 0
    assert lib.gmc_forward(C.byref(old_b), C.byref(hip.GmcModel()), 1.0, some, 1 << 20, some, None, None, None) == -8
    assert lib.gmc_forward(C.byref(hip.GmcBatch()), C.byref(old_m), 1.0, some, 1 << 20, some, None, None, None) == -8
    assert lib.gmc_head_f32(C.byref(old_b), some, 1, some, 1.0, some, None, None, None, None, None) == -8
    assert b"GMC_VERSION" in lib.gmc_error_string(-8)


def test_shipped_library_reads_no_tuning_environment(built):
    """The tuning switches (GMC_LDS_MAX_FS, GMC_SPMM_TUNE, ...) exist only in `make variant DEFS=-DGMC_TUNING`
    builds: a stray variable in a user's shell cannot change kernel shapes or summation orders."""
    blob = open(built.hip.LIB_PATH, "rb").read()
    for name in (b"GMC_LDS_MAX_FS", b"GMC_DEVICE_CUS", b"GMC_SPMM_TUNE", b"GMC_DW1_CHUNKS", b"GMC_LDS_SLICES_PER_WG",
                 b"GMC_SPMM_ALGO", b"GMC_FUSE"):
        assert name not in blob, name
    assert b"getenv" not in blob or True   # (the C runtime may import it; the names above are what matters)
