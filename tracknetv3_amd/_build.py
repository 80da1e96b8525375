"""Build libtnv3_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and on first import
when the library is missing but hipcc is available.

csrc/tnv3_capi.hip is compiled once per kernel family (-DTNV3_TU_<FAMILY>), in parallel, into build/*.o; a family is
recompiled only when one of the files its depfile lists changed.  `build_diag()` makes libtnv3_diag.so (measurement twins,
include/tracknetv3_hip_diag.h) from the same source; the product never loads it."""
import os
import shutil  # noqa: I001

import torch  # noqa: F401  (loaded first so that libtnv3_hip.so binds to torch's HIP runtime, not a second copy)
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "tnv3_capi.hip")
LIB = os.path.join(PKG_DIR, "libtnv3_hip.so")
DIAG_LIB = os.path.join(PKG_DIR, "libtnv3_diag.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")
FAMILIES = ("MISC", "CONV", "WINO", "WINO43", "UP2X", "WGRAD", "TRAIN")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-cuda-compat"]
# Per-family extras.  The Winograd kernels are compiled without the SLP vectoriser: it pairs the fp32 adds of the patch / output
# transforms into v_pk_add_f32 at the price of a v_mov per operand pair (61 instead of 56 vector instructions per chunk) and of
# ~40 temporaries in the output transform, which pushed the persistent kernel past its 256-register budget (26 spills).
FAMILY_FLAGS = {"WINO": ["-fno-slp-vectorize"], "WINO43": ["-fno-slp-vectorize"], "DIAG": ["-fno-slp-vectorize"]}


def _sources():
    out = [SRC, os.path.join(PKG_DIR, "..", "include", "tracknetv3_hip.h"), os.path.join(PKG_DIR, "..", "include", "tracknetv3_hip_diag.h")]
    for root, _, files in os.walk(os.path.join(PKG_DIR, "csrc")):
        out += [os.path.join(root, f) for f in files]
    return out


def source_sha256():
    """sha256 over what the library is built FROM: every file under csrc/ and the two C headers (relative path + bytes, sorted) and the compile
    flags.  Unlike the sha256 of libtnv3_hip.so -- which embeds the build directory's path, so the same sources hash differently under /tmp
    than under /root/repo -- this key is the same wherever the tree is built: bench.py replays profiled counters (profiles/conv_traffic.json)
    for a build with the same sources, the profile scripts record it next to the library's own hash."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(PKG_DIR)
    for path in sorted(os.path.realpath(p_) for p_ in _sources() if os.path.isfile(p_)):
        h.update(os.path.relpath(path, os.path.realpath(root)).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, sorted(FAMILY_FLAGS.items()), FAMILIES)).encode())
    return h.hexdigest()


def _newer_than(target, paths):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in paths if os.path.exists(s))


def is_stale():
    return _newer_than(LIB, _sources())


def _dep_files(depfile):
    """Prerequisites listed in a make-style depfile (hipcc -MD), or None when unreadable."""
    try:
        with open(depfile) as f:
            text = f.read().replace("\\\n", " ")
    except OSError:
        return None
    deps = []
    for line in text.splitlines():
        if ":" in line:
            deps += line.split(":", 1)[1].split()
    return [d for d in deps if d.startswith(os.path.dirname(PKG_DIR))] or None


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libtnv3_hip.so")
    return hipcc


def _compile(hipcc, tag, defines, verbose):
    obj = os.path.join(OBJ_DIR, f"tnv3_{tag.lower()}.o")
    dep = obj[:-2] + ".d"
    deps = _dep_files(dep)
    if os.path.exists(obj) and deps is not None and not _newer_than(obj, deps + [__file__]):
        return obj, False
    cmd = [hipcc] + FLAGS + FAMILY_FLAGS.get(tag, []) + [f"-D{d}" for d in defines] + ["-MD", "-MF", dep, "-c", SRC, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(obj + ".tmp", obj)
    return obj, True


def _link(hipcc, objs, lib, verbose):
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)


def build(force=False, verbose=False):
    """Compile every HIP kernel + the C ABI for gfx950.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            if f.startswith("tnv3_") and not f.startswith("tnv3_diag"):
                os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(len(FAMILIES), os.cpu_count() or 1)) as pool:
        res = list(pool.map(lambda fam: _compile(hipcc, fam, [f"TNV3_TU_{fam}"], verbose), FAMILIES))
    _link(hipcc, [o for o, _ in res], LIB, verbose)
    return LIB


def build_diag(force=False, verbose=False):
    """libtnv3_diag.so: the timing twins / MFMA probe (a measurement tool for scripts/, never loaded by the package)."""
    if not force and not _newer_than(DIAG_LIB, _sources()):
        return DIAG_LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    obj, _ = _compile(hipcc, "DIAG", ["TNV3_DIAG", "TNV3_TU_DIAG"], verbose)
    _link(hipcc, [obj], DIAG_LIB, verbose)
    return DIAG_LIB
