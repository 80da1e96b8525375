"""Build libtnv3_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and on first import
when the library is missing but hipcc is available."""
import os
import shutil  # noqa: I001

import torch  # noqa: F401  (loaded first so that libtnv3_hip.so binds to torch's HIP runtime, not a second copy)
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "tnv3_capi.hip")
LIB = os.path.join(PKG_DIR, "libtnv3_hip.so")


def _sources():
    out = [SRC, os.path.join(PKG_DIR, "..", "include", "tracknetv3_hip.h")]
    for root, _, files in os.walk(os.path.join(PKG_DIR, "csrc")):
        out += [os.path.join(root, f) for f in files]
    return out


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources() if os.path.exists(s))


def build(force=False, verbose=False):
    """Compile every HIP kernel + the C ABI for gfx950.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libtnv3_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           SRC, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB
