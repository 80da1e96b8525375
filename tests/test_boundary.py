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
    assert lib.tnv3_abi_version() == 1
    lib.tnv3_conv3x3_num_configs.restype = ctypes.c_int
    assert lib.tnv3_conv3x3_num_configs() >= 1


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
