import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libtnv3_emu.so")
HOST_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _emu_sources():
    out = [os.path.join(EMU_DIR, "hip_emu.h"), os.path.join(EMU_DIR, "emu_api.cpp"),
           os.path.join(ROOT, "include", "tracknetv3_hip.h")]
    for root, _, files in os.walk(os.path.join(ROOT, "tracknetv3_amd", "csrc")):
        out += [os.path.join(root, f) for f in files]
    return out


def build_emulator():
    """Compile the UNCHANGED kernel headers + dispatch code for the host SIMT emulator (test tool)."""
    if os.path.exists(EMU_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(EMU_LIB) for s in _emu_sources() + [os.path.abspath(__file__)]):
        return EMU_LIB
    cxx = HOST_CLANG if os.path.exists(HOST_CLANG) else "clang++"
    # -DTNV3_DIAG: the test tool dispatches the measurement twins of libtnv3_diag.so as well (they are the bit-identical references of several
    # kernel tests); that the PRODUCT build refuses them is tests/test_kernel_resources.py's and tests/test_gpu_tracknet.py's business
    cmd = [cxx, "-std=c++17", "-O2", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-psabi", "-DTNV3_DIAG",
           "-include", os.path.join(EMU_DIR, "hip_emu.h"), os.path.join(EMU_DIR, "emu_api.cpp"), "-o", EMU_LIB + ".tmp"]
    subprocess.run(cmd, check=True)
    os.replace(EMU_LIB + ".tmp", EMU_LIB)
    return EMU_LIB


@pytest.fixture(scope="session")
def emu_lib_path():
    import shutil
    if not os.path.exists(HOST_CLANG) and shutil.which("clang++") is None:
        pytest.skip("no host clang++ for the SIMT emulator")
    return build_emulator()          # a compile error of the kernel headers must FAIL the suite, not skip it


@pytest.fixture()
def emu(emu_lib_path):
    """Bind the product's ctypes layer to the emulator library for one test, then unbind."""
    from tracknetv3_amd import _lib
    _lib.use_library(emu_lib_path)
    yield _lib
    _lib.reset_library()


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from tracknetv3_amd import _lib
    _lib.reset_library()
    _lib.load()          # must be the real HIP library; raises loudly otherwise
    assert not _lib.is_emulator()
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 0], ids=["f43fwd", "f22fwd"])
def train_fwd(request, monkeypatch):
    """Both training forwards: the F(4x4) kernel's statistics epilogue (tuning.WINO43_TRAIN, the default) and the F(2x2) kernels'."""
    from tracknetv3_amd import tuning
    monkeypatch.setattr(tuning, "WINO43_TRAIN", bool(request.param))
    return int(request.param)


def write_report(name, obj):
    """Measurement reports of the parity tests: written only when TNV3_REPORT_DIR names a directory (scripts/gpu_session.sh sets it)."""
    import json
    out = os.environ.get("TNV3_REPORT_DIR")
    if not out:
        return
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump(obj, f, indent=1)
