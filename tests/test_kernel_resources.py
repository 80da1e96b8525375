"""CPU: register / occupancy budget of the hot kernels, read from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
A runtime flag added to the conv kernel once cost 15 VGPRs = one resident workgroup per CU (-4 % end to end) without any
test noticing; this pins the budgets the tuned table relies on."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from concurrent.futures import ThreadPoolExecutor
    from tracknetv3_amd import _build
    tmp = tmp_path_factory.mktemp("res")

    def one(fam):                                     # same per-family flags as the product build (tracknetv3_amd/_build.py)
        return subprocess.run([hipcc] + [f for f in _build.FLAGS if f != "-fPIC"] + _build.FAMILY_FLAGS.get(fam, []) +
                              ["-S", "--cuda-device-only", f"-DTNV3_TU_{fam}", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp / f"{fam}.s"),
                               os.path.join(ROOT, "tracknetv3_amd", "csrc", "tnv3_capi.hip")], capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(len(_build.FAMILIES), os.cpu_count() or 1)) as pool:
        runs = list(pool.map(one, _build.FAMILIES))
    kernels, cur = {}, None
    for r in runs:
        assert r.returncode == 0, r.stderr[-2000:]
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
                kernels[cur] = {}
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
            if m and cur:
                kernels[cur][m.group(1).strip()] = int(m.group(2))
    assert len(kernels) > 30
    kernels["__asm__"] = [str(tmp / f"{fam}.s") for fam in _build.FAMILIES]
    return kernels


def _find(kernels, *needles):
    hits = [v for k, v in kernels.items() if all(n in k for n in needles)]
    assert len(hits) == 1, (needles, len(hits))
    return hits[0]


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _split_ops(text):
    return [o.strip() for o in re.split(r",\s*(?![^\[]*\])", text)]


def store_data_hazards(path, need=2):
    """(line, store, clobbering instruction) of every >8-byte vector store whose data registers a VALU instruction writes within `need`
    wait states.  gfx940+ needs two; hipcc's hazard recogniser skips buffer stores with an SGPR soffset (their extra issue cycle covers
    one), and the F(4x4) kernel's statistics epilogue lost elements 0, 1 of a float4 to a v_add_f64 that way on an MI355X."""
    lines = [l.split(";")[0].strip() for l in open(path)]
    ins = [(i, l) for i, l in enumerate(lines) if l and not l.endswith(":") and not l.startswith(".")]
    hits = []
    for k, (i, l) in enumerate(ins):
        m = re.match(r"((?:buffer|global|flat|scratch)_store_dwordx[34])\s+(.*)", l)
        if not m:
            continue
        ops = _split_ops(m.group(2))
        data = _vregs(ops[0]) if m.group(1).startswith("buffer") else _vregs(ops[1])
        ws = 0
        for (_, n) in ins[k + 1:k + 1 + need]:
            if ws >= need:
                break
            mm = re.match(r"s_nop (\d+)", n)
            if mm:
                ws += int(mm.group(1)) + 1
                continue
            if n.startswith("v_") and _vregs(_split_ops(n.split(None, 1)[1])[0]) & data:
                hits.append((i + 1, l, n))
                break
            ws += 1
    return hits


def test_no_valu_write_of_store_data_within_two_wait_states(resources):
    hits = [(os.path.basename(p), h) for p in resources["__asm__"] for h in store_data_hazards(p)]
    assert not hits, hits[:5]


def test_no_vgpr_spills_anywhere(resources):
    resources = {k: v for k, v in resources.items() if k != "__asm__"}
    bad = {k: v for k, v in resources.items() if v.get("VGPRs Spill", 0) or v.get("ScratchSize [bytes/lane]", 0)}
    assert not bad, list(bad)


def test_product_library_carries_no_measurement_twins(resources):
    """what the product build (no -DTNV3_DIAG) instantiates.  Gone with ABI 5 / 6: the one-wave and xi-split F(2x2) kernels, the
    64-channel form of the 128-channel F(2x2) kernel (variant 7), the 32x32x2 F(4x4) kernel, its pack kernel, the Winograd weight-gradient
    generations 0 / 2 / 3 / 4 / 6 / 7 and the LDS-DMA staged direct weight gradient.  (tests/test_gpu_tracknet.py checks the refusals at run time.)"""
    names = [n for n in resources if n != "__asm__"]
    for gone in ("conv3x3_wino_mfma_kernelINS_7WinoCfg", "conv3x3_wino_split_mfma_kernel", "22conv3x3_wino43_kernelI", "conv3x3_wino43_pack_kernel",
                 "wgrad3x3_dma_kernel", "wgrad_wino3_mfma_kernel", "20wgrad_wino_mfma_kernelE", "WinoV6CfgILi0ELi1ELi0ELi1ELi0ELi2E"):
        assert not any(gone in n for n in names), (gone, [n for n in names if gone in n])
    for kept in ("conv3x3_wino43s_kernelILi8ELi0E", "conv3x3_wino43s_kernelILi4ELi0E", "conv3x3_wino_stream_mfma_kernel", "conv3x3_wino_a128_stream_kernel",
                 "wgrad_wino43_kernel", "wgrad_up2x_wino43_kernel", "wgrad_wino2_mfma_kernelILi0E", "wgrad_wino5_mfma_kernel", "wgrad3x3_mfma_kernel"):
        assert any(kept in n for n in names), kept


def test_conv_and_wgrad_budgets(resources):
    cfg = "conv3x3_mfma_kernelINS_7ConvCfgI"
    # (template arguments, minimum waves per SIMD) of the configurations the tuned table uses
    for args, occ in (("Li2ELi1ELi1ELi8ELi8ELi32ELi8ELi1ELi2ELi1ELi0ELi0E", 4),     # cfg 10
                      ("Li2ELi1ELi2ELi4ELi4ELi32ELi4ELi1ELi2ELi1ELi0ELi0E", 6),     # cfg 11
                      ("Li2ELi1ELi1ELi4ELi4ELi32ELi8ELi1ELi2ELi1ELi0ELi0E", 3),     # cfg 12
                      ("Li2ELi1ELi1ELi4ELi4ELi32ELi8ELi1ELi1ELi0ELi0ELi0E", 3),     # cfg 7
                      ("Li2ELi1ELi1ELi8ELi8ELi32ELi8ELi1ELi1ELi0ELi0ELi0E", 4),     # cfg 8
                      ("Li2ELi1ELi2ELi4ELi4ELi32ELi4ELi1ELi1ELi0ELi0ELi0E", 6)):    # cfg 9
        k = _find(resources, cfg + args)
        assert k["Occupancy [waves/SIMD]"] >= occ, (args, k)
    for args in ("WgradCfgILi4ELi1ELi4ELi32ELi9E", "WgradCfgILi2ELi2ELi4ELi32ELi9E", "WgradCfgILi2ELi2ELi4ELi32ELi4E"):
        k = _find(resources, "wgrad3x3_mfma_kernel", args)
        assert k["VGPRs"] + k.get("AGPRs", 0) <= 512 and k["Occupancy [waves/SIMD]"] >= 1
    k = _find(resources, "wgrad3x3_mfma_kernel", "WgradCfgILi4ELi2ELi4ELi32ELi4E")        # 2x2-window, 8 waves: two per SIMD
    assert k["Occupancy [waves/SIMD]"] >= 2 and k["VGPRs Spill"] == 0
    # (the one-wave and xi-split F(2x2) generations -- variants 0 / 2 / 4 -- and the weight-gradient generations 0 / 3 / 4 / 6 / 7 left the product
    #  library with ABI 5: libtnv3_diag.so only)
    assert not any("conv3x3_wino_mfma_kernelINS_7WinoCfg" in n or "conv3x3_wino_split_mfma_kernel" in n for n in resources)
    # variants 3 and 4 (one tile per workgroup), 5 (the streaming persistent kernel) and 6 (its 128-channel form with the filter operand
    # in registers -- which spilled 700+ registers until the two groups' programs were predicated instead of branched): the same budget
    for name in ("conv3x3_wino_v3_mfma_kernelINS_9WinoV3CfgILi8ELi0ELi0ELi0ELi0ELi0ELi0ELi0E",
                 "conv3x3_wino_stream_mfma_kernelINS_9WinoV3CfgILi8ELi0ELi0ELi0ELi0ELi0ELi0ELi1E", "conv3x3_wino_a128_stream_kernelINS_9WinoV6CfgILi0ELi1ELi0ELi1ELi0ELi1E"):
        k = _find(resources, name)
        assert k["Occupancy [waves/SIMD]"] >= 2 and k["VGPRs"] + k.get("AGPRs", 0) <= 256 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, k
        assert k["LDS Size [bytes/block]"] <= 160 * 1024
    # Winograd weight gradient, two waves per SIMD: the role-split generations and the production kernel (every wave streams and transforms)
    for name in ("wgrad_wino2_mfma_kernelILi0E", "wgrad_wino5_mfma_kernelINS_13WgradWino5CfgILi3ELi0E"):
        k = _find(resources, name)
        assert k["Occupancy [waves/SIMD]"] >= 2 and k["VGPRs"] + k.get("AGPRs", 0) <= 256 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, k
        assert k["LDS Size [bytes/block]"] <= 160 * 1024
    # the 16x16x4 F(4x4) kernel (variant 0), both geometries, plain / statistics epilogue: 144 accumulators + named filter quads + the patch
    # transform in 256 registers, two waves per SIMD, no spill traffic (its steps end in a COUNTED vmcnt)
    # (every instantiation: geometry 4 / 8 x plain / statistics / pooled second output / the upsampled halves' 25-product forward (MODE 1) and
    #  data gradient (MODE 2), the latter also with the BatchNorm-backward sums epilogue)
    w43s = {n: k for n, k in resources.items() if "conv3x3_wino43s_kernelILi" in n}
    assert len(w43s) >= 10 and sum("ELi0ELi0ELi0ELi2E" in n for n in w43s) == 4 and all(any(f"conv3x3_wino43s_kernelILi{c}ELi{st}E" in n for n in w43s) for c in (4, 8) for st in (0, 1)), sorted(w43s)
    for name, k in w43s.items():
        assert k["Occupancy [waves/SIMD]"] >= 2 and k["VGPRs"] + k.get("AGPRs", 0) <= 256 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, (name, k)
        assert k["LDS Size [bytes/block]"] <= 160 * 1024
    # the F(4x4) weight gradient: 144 accumulators + both operand transforms, two waves per SIMD, its strips end in counted vmcnt waits
    k = _find(resources, "wgrad_wino43_kernel")
    assert k["Occupancy [waves/SIMD]"] >= 2 and k["VGPRs"] + k.get("AGPRs", 0) <= 256 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, k
    assert k["LDS Size [bytes/block]"] <= 160 * 1024
    # the upsampled halves' 25-of-36 weight gradient (round 5): 100 accumulators, two waves per SIMD
    k = _find(resources, "wgrad_up2x_wino43_kernel")
    assert k["Occupancy [waves/SIMD]"] >= 2 and k["VGPRs"] + k.get("AGPRs", 0) <= 256 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, k
    assert k["LDS Size [bytes/block]"] <= 160 * 1024
    # (the 32x32x2 predecessor conv3x3_wino43_kernel is a twin of libtnv3_diag.so since ABI 6: test_product_library_carries_no_measurement_twins)
    for name, occ in (("conv_up2x_mfma_kernelINS_11ConvUp2xCfgILi2ELi2ELi4ELi2ELi4E", 4), ("conv_up2x_mfma_kernelINS_11ConvUp2xCfgILi2ELi1ELi4ELi2ELi4E", 4),
                      ("dgrad_up2x_mfma_kernelINS_12DgradUp2xCfgILi2ELi2ELi4ELi1ELi2E", 4)):
        k = _find(resources, name)
        assert k["Occupancy [waves/SIMD]"] >= occ and k["VGPRs Spill"] == 0, (name, k)


def test_store_data_hazard_scanner_on_known_patterns(tmp_path):
    """The scanner itself: the sequence that corrupted stores on the MI355X is reported, covered forms are not."""
    cases = {
        # (a) what hipcc emitted for the F(4x4) statistics epilogue: SGPR offset, VALU write of v[166:167] in the next instruction
        "bad_next": ("buffer_store_dwordx4 v[166:169], v226, s[56:59], s8 offen\nv_add_f64 v[166:167], v[162:163], v[164:165]\n", 1),
        # (b) one unrelated instruction in between is one wait state: still one short of two
        "bad_one_between": ("buffer_store_dwordx4 v[146:149], v226, s[56:59], s6 offen\ns_mul_i32 s6, s74, 12\nv_cvt_f64_f32_e32 v[146:147], v147\n", 1),
        "ok_two_between": ("buffer_store_dwordx4 v[146:149], v226, s[56:59], s6 offen\ns_mul_i32 s6, s74, 12\ns_mul_i32 s7, s74, 12\nv_mov_b32_e32 v146, 0\n", 0),
        "ok_nop": ("buffer_store_dwordx4 v[10:13], v1, s[4:7], s8 offen\ns_nop 1\nv_mov_b32_e32 v10, 0\n", 0),
        "ok_other_register": ("global_store_dwordx4 v[6:7], v[2:5], off\nv_mov_b32_e32 v6, 0\n", 0),      # the ADDRESS registers may be rewritten
        "bad_global": ("global_store_dwordx4 v[6:7], v[2:5], off\nv_mov_b32_e32 v3, 0\n", 1),
        "ok_8_bytes": ("buffer_store_dwordx2 v[2:3], v1, s[4:7], s8 offen\nv_mov_b32_e32 v2, 0\n", 0),
    }
    for name, (asm, want) in cases.items():
        p = tmp_path / f"{name}.s"
        p.write_text(asm)
        assert len(store_data_hazards(str(p))) == want, name
