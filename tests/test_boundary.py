"""CPU: the drop-in boundary -- C-ABI exports, loader behaviour, model factory, state_dict layout, checkpoints."""
import ctypes
import io
import os
import re

import pytest
import torch

from conftest import ROOT
from oracle import nets


def test_capi_library_builds_loads_and_exports_every_declared_symbol():
    from tracknetv3_amd import _build
    path = _build.build()                      # hipcc cross-compiles gfx950 without a GPU
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "tracknetv3_hip.h")).read()
    declared = set(re.findall(r"\b(tnv3_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 10
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tracknetv3_hip.h but not exported"
    lib.tnv3_abi_version.restype = ctypes.c_int
    assert lib.tnv3_abi_version() == 8
    lib.tnv3_conv3x3_num_configs.restype = ctypes.c_int
    assert lib.tnv3_conv3x3_num_configs() >= 1
    from tracknetv3_amd import _lib
    assert set(_lib.EXPORTS) == declared, set(_lib.EXPORTS) ^ declared      # the ctypes layer binds exactly the header


def test_abi_has_no_process_wide_state_and_no_wrong_result_kernels():
    """'no mutable global state'.  Kernel families are per-call arguments; the timing twins (wrong results by
    design) and the MFMA probe live in libtnv3_diag.so, which the product never loads."""
    import subprocess
    from tracknetv3_amd import _build
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tracknetv3_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(tnv3_[a-z0-9_]+)\s*\(", header))
    # every entry point either takes a stream (enqueues work) or is a pure size / capability query
    for name in declared:
        proto = re.search(r"\b" + name + r"\s*\(([^;]*)\);", header, re.S).group(1)
        is_query = name.endswith(("_bytes", "_floats", "_supported", "_num_configs", "_config_info", "_layout", "_has_stats", "_pick", "_stats_tiles", "abi_version", "last_error"))
        assert ("tnv3_stream_t" in proto) != is_query, name
    assert not [n for n in declared if n.endswith("_variant") or "diag" in n or "probe" in n]
    exported = subprocess.run(["nm", "-D", "--defined-only", _build.build()], capture_output=True, text=True, check=True).stdout
    names = set(re.findall(r" [TW] (tnv3_\w+)", exported))
    assert names == declared, names ^ declared
    # the diag library builds from the same sources and exports only tnv3_diag_*
    dpath = _build.build_diag()
    dnames = set(re.findall(r" [TW] (tnv3_\w+)", subprocess.run(["nm", "-D", "--defined-only", dpath], capture_output=True, text=True,
                                                                    check=True).stdout))
    dheader = open(os.path.join(ROOT, "include", "tracknetv3_hip_diag.h")).read()
    assert dnames == set(re.findall(r"\b(tnv3_diag_[a-z0-9_]+)\s*\(", dheader)), dnames
    # nothing under tracknetv3_amd/ mentions the diag library except the build script
    for root, _, files in os.walk(os.path.join(ROOT, "tracknetv3_amd")):
        for f in files:
            if f.endswith(".py") and f != "_build.py":
                assert "tnv3_diag" not in open(os.path.join(root, f)).read(), f


def test_product_path_has_no_cpu_fallback():
    from tracknetv3_amd import _lib, ops
    _lib.reset_library()
    _lib.load()
    assert not _lib.is_emulator()
    with pytest.raises(_lib.Tnv3Error, match="no CPU fallback"):
        ops.maxpool2x2(torch.zeros(1, 1, 4, 8))
    from tracknetv3_amd.utils.general import get_model
    m = get_model("TrackNet", 3, "").eval()
    with pytest.raises(_lib.Tnv3Error):
        m(torch.zeros(1, 9, 16, 32))


def test_get_model_contract():
    from tracknetv3_amd.utils.general import get_model, HEIGHT, WIDTH, COOR_TH
    from tracknetv3_amd.model import TrackNet, InpaintNet
    for bg in ("", "subtract", "subtract_concat", "concat", None, "zzz"):
        for L in (1, 3, 8):
            m = get_model("TrackNet", L, bg)
            assert isinstance(m, TrackNet) and (m.in_dim, m.out_dim) == nets.tracknet_dims(L, bg)
    assert isinstance(get_model("InpaintNet"), InpaintNet)
    with pytest.raises(ValueError, match="Invalid model name."):
        get_model("Nope")
    assert (HEIGHT, WIDTH) == (288, 512) and abs(COOR_TH - 50 / (288 ** 2 + 512 ** 2) ** 0.5) < 1e-15


def test_state_dict_layout_and_checkpoint_roundtrip():
    from tracknetv3_amd.utils.general import get_model
    m = get_model("TrackNet", 8, "concat")
    shapes = nets.tracknet_state_shapes(27, 8)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and len(sd) == 104
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k][0]) and v.dtype == shapes[k][1], k
    assert sum(p.numel() for p in m.parameters()) == 11341000
    assert len(list(m.parameters())) == 53 and all(p.is_leaf for p in m.parameters())
    ip = get_model("InpaintNet")
    ishapes = nets.inpaintnet_state_shapes()
    assert list(ip.state_dict().keys()) == list(ishapes.keys()) and len(ishapes) == 18
    assert sum(p.numel() for p in ip.parameters()) == 520610
    # reference checkpoint dict layout (train.py:283-301) survives torch.save / weights_only load
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ckpt = dict(epoch=3, max_val_acc=0.5, model=m.state_dict(), optimizer=opt.state_dict(), scheduler=None,
                param_dict=dict(model_name="TrackNet", seq_len=8, bg_mode="concat"))
    buf = io.BytesIO()
    torch.save(ckpt, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=True)
    m2 = get_model(back["param_dict"]["model_name"], back["param_dict"]["seq_len"], back["param_dict"]["bg_mode"])
    assert m2.load_state_dict(back["model"], strict=True).missing_keys == []
    # synthetic reference-layout state loads too
    m2.load_state_dict(nets.synth_state(shapes, 5), strict=True)


def test_infer_split_gate_is_thread_local_and_restored():
    """model.no_infer_split(): nests, restores the previous state, and does not leak into other threads."""
    import threading
    from tracknetv3_amd import model as M
    assert not M._NO_SPLIT.active
    seen = []
    with M.no_infer_split():
        assert M._NO_SPLIT.active
        with M.no_infer_split():
            assert M._NO_SPLIT.active
        assert M._NO_SPLIT.active
        t = threading.Thread(target=lambda: seen.append(M._NO_SPLIT.active))
        t.start()
        t.join()
    assert not M._NO_SPLIT.active and seen == [False]
