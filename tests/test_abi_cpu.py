"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the
operator modules keep the reference's state_dict layout, the product path refuses CPU tensors."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import Fixture
from mpnn_amd import _lib, synth


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.load()._name)
    names = _lib.declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), "libmpnn_amd.so lacks %s declared in include/mpnn_amd.h" % n
    assert _lib.load().mpnn_version() >= 100


def test_binding_is_derived_from_the_header():
    """The ctypes prototypes come from include/mpnn_amd.h itself (no second, hand-kept table): every pointer is an
    address, scalars keep their C width, and the kernel entry points end with the stream."""
    sigs = _lib.header_signatures()
    assert len(sigs) >= 30
    assert sigs["mpnn_last_error_string"] == (ctypes.c_char_p, [])
    assert sigs["mpnn_csr_workspace_bytes"] == (ctypes.c_size_t, [ctypes.c_int64])
    res, args = sigs["mpnn_segsum_f32"]                      # (msg, row_ptr, w, out, int64 rows, int F, stream)
    assert res is ctypes.c_int and args == [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    res, args = sigs["mpnn_masked_bn_fwd_f32"]
    assert args.count(ctypes.c_float) == 1 and args[-3:] == [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    for name, (res, args) in sigs.items():
        if name.endswith("_f32"):
            assert res is ctypes.c_int and args[-1] is ctypes.c_void_p, name
    lib = _lib.load()
    for name, (res, args) in sigs.items():                   # what the loaded functions were given
        fn = getattr(lib, name)
        assert fn.restype is res and list(fn.argtypes) == args, name


def test_no_cpu_fallback():
    from mpnn_amd import ops
    with pytest.raises(_lib.MpnnError):
        ops.segsum_raw(torch.zeros(4, 8), torch.zeros(3, dtype=torch.int32), None, 2)
    from mpnn_amd.mpnn_functions import GRUUpdate
    g = GRUUpdate(8, 8)
    with pytest.raises(_lib.MpnnError):
        g(torch.zeros(2, 3, 8), torch.zeros(2, 3, 8), torch.ones(2, 3, 1))


def test_rejected_calls_report_an_error_string():
    lib = _lib.load()
    rc = lib.mpnn_segsum_f32(None, None, None, None, 4, 8, None)
    assert rc == -1 and b"mpnn_segsum_f32" in lib.mpnn_last_error_string()
    rc = lib.mpnn_gru_update_f32(None, None, None, None, None, None, None, None, None, None, 0, 4, 100000, None)
    assert rc == -1
    assert lib.mpnn_edge_message_f32(None, None, None, None, None, None, None, 0, 0, 0, 8, 8, None) == 0   # empty


@pytest.mark.parametrize("name,prefix", [("model_basic_h8", ""), ("model_lipo_T3_train", "")])
def test_state_dict_layout_matches_reference(name, prefix):
    f = Fixture(name)
    if name.startswith("model_basic"):
        from mpnn_amd.models.basic_model import BasicModel
        from mpnn_amd.models.graph_model_wrapper import GraphWrapper
        m = GraphWrapper(BasicModel(8, 4, 8, 9, 6, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}))
    else:
        from mpnn_amd.models.lipo_basic_model import BasicModel
        from mpnn_amd.models.graph_norm_wrapper import GraphWrapper
        m = GraphWrapper(BasicModel(22, 7, 22, 9, 38, message_opts={}, agg_opts={}, update_opts={},
                                    readout_opts={}, message_steps=3), 3)
    sd = m.state_dict()
    assert set(sd) == set(f.params), set(sd) ^ set(f.params)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(f.params[k].shape), k
    m.load_state_dict(f.params)     # reference checkpoints load as they are
    # the 50 tower aliases are ONE tensor here as well
    import re
    tower = [k for k in sd if re.search(r"\.edge_map\.\d+\.0\.weight$", k)]
    assert len(tower) == 50 and len({sd[k].data_ptr() for k in tower}) == 1


def test_mutable_default_opts_are_shared_like_the_reference():
    from mpnn_amd.models.basic_model import BasicModel
    BasicModel(8, 4, 8, 9, 6)
    assert BasicModel.__init__.__defaults__[1]["node_features"] == 8    # reference quirk


def test_synthetic_molecules():
    mb = synth.make_molecules(2000, 16, seed=3)
    V, E = mb.num_atoms, mb.num_edges
    deg = np.diff(mb.row_ptr)
    assert mb.row_ptr[-1] == E and deg.sum() == E
    dst = np.repeat(np.arange(V), deg)
    mol = np.repeat(np.arange(mb.num_mols), mb.n_atoms)
    assert (mol[dst] == mol[mb.col_idx]).all()                   # no cross-molecule edges
    assert (dst != mb.col_idx).all()                             # no self loops
    key = dst.astype(np.int64) * V + mb.col_idx
    assert (np.diff(key) > 0).all()                              # sorted by (dst, src), no duplicates
    rev = mb.col_idx.astype(np.int64) * V + dst
    assert np.array_equal(np.sort(rev), key)                     # symmetric
    order = np.argsort(rev, kind="stable")
    assert np.array_equal(mb.bond_type[order], mb.bond_type)     # same bond type both directions
    assert 55 < E / mb.num_mols < 65
    sk = synth.make_molecules(500, 4, seed=5, dist="skewed")
    assert np.diff(sk.row_ptr).max() > 12                        # heavy-tailed hubs
    d = synth.to_dense(synth.select(mb, [3, 1, 4]))
    assert d["adj"].shape[0] == 3 and (d["adj"] == d["adj"].transpose(0, 2, 1)).all()


def test_reference_module_names_importable_with_package_dir_on_path():
    """INTEGRATION.md level 1: with mpnn_amd/ on sys.path the reference's own import lines work."""
    import subprocess
    import sys
    from conftest import REPO
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from mpnn_functions import *\n"
            "from mpnn_functions.message.ggnn_msg_pass import GGNNMsgPass\n"
            "from models.basic_model import BasicModel\n"
            "from models.graph_model_wrapper import GraphWrapper\n"
            "from models.lipo_basic_model import BasicModel as Lipo\n"
            "from models.mask_batch_norm import MaskBatchNorm1d\n"
            "m = GraphWrapper(BasicModel(8, 4, 8, 9, 6, message_func=EdgeNetwork, message_agg_func=AdjMsgAgg,"
            " update_func=GRUUpdate, readout_func=GraphLevelOutput))\n"
            "print(len(m.state_dict()))\n") % (REPO + "/mpnn_amd", REPO)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "63"


def test_mfma_pricing_follows_the_math_switch(monkeypatch):
    """bench.py prices each contraction kernel against the 16-bit MFMA peak divided by the MFMAs it issues per fp32
    product (ops.mfma_per_product): six for the three-way bf16 splits of the per-edge message kernels, three for the
    kernels on two fp16 pieces, zero (= the fp32 pipe itself) under MPNN_GRU_MATH=fp32 -- the one math switch the library
    reads."""
    from mpnn_amd import ops
    monkeypatch.delenv("MPNN_GRU_MATH", raising=False)
    assert [ops.mfma_per_product("gru_update_bwd", h) for h in (64, 128, 256)] == [3, 3, 3]
    assert [ops.mfma_per_product("gru_update", h) for h in (64, 128, 256)] == [3, 3, 3]
    assert ops.mfma_per_product("message_aggregate", 64) == 3 and ops.mfma_per_product("edge_message", 128) == 6
    monkeypatch.setenv("MPNN_GRU_MATH", "fp32")
    assert ops.mfma_per_product("gru_update_bwd", 64) == 0 and "fp32 matrix pipe" in ops.math_description()


def test_a_library_not_built_from_this_tree_is_never_loaded_silently(monkeypatch):
    """VERDICT r3 / ADVICE r3: load() used an existing .so whatever it was built from.  The manifest next to the library
    holds the hash of sources + headers + flags; a mismatch makes the library stale, and load() then rebuilds or raises."""
    import os
    import subprocess
    import sys
    from mpnn_amd import build
    _lib.load()
    assert not build._stale()
    monkeypatch.setattr(build, "source_hash", lambda: "an edited source")
    assert build._stale()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "build", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hipcc not found")))
    with pytest.raises(_lib.MpnnError):
        _lib.load()
    # experiment flags never land on the product library's path
    out = subprocess.run([sys.executable, "-c", "from mpnn_amd import build; print(build.LIB)"], capture_output=True, text=True,
                         env=dict(os.environ, MPNN_EXTRA_HIPCC_FLAGS="-DMPNN_ABL_HOT_ROWS"), cwd=os.path.dirname(build.HERE))
    assert "variant_" in out.stdout and out.stdout.strip() != build.LIB
