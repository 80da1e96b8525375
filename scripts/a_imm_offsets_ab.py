"""A/B of the F(4x4) kernel's filter-operand addressing (round 6): one scalar offset per quad (36 scalar adds per step: the product) against
the quads' (q & 3) * 1024 bytes in the load's immediate offset (-DTNV3_A_IMM_OFFSETS=1: 9 adds per step) on TrackNet's plain-layer shapes
at batch 10, eval-mode epilogue, both geometries.  Result (profiles/r06_a_imm_offsets_ab.json, recorded when the immediate form was the
product build and the scalar one the alternative): 0.0 to +2 % per launch for the immediate form -- not adopted.
  python scripts/a_imm_offsets_ab.py build      (in the build container: writes scripts/libtnv3_hip_alt_a_offsets.so with the immediate form)
  python scripts/a_imm_offsets_ab.py            (on the GPU box)"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
ALT = os.path.join(HERE, "libtnv3_hip_alt_a_offsets.so")
SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128), (256, 512, 36, 64),
          (512, 512, 36, 64))


def build():
    from tracknetv3_amd import _build
    hipcc = _build._hipcc()
    objs = []
    os.makedirs("/tmp/alt_build", exist_ok=True)
    procs = []
    for fam in _build.FAMILIES:
        obj = f"/tmp/alt_build/tnv3_{fam.lower()}.o"
        cmd = [hipcc] + _build.FLAGS + _build.FAMILY_FLAGS.get(fam, []) + [f"-DTNV3_TU_{fam}", "-DTNV3_A_IMM_OFFSETS=1", "-c", _build.SRC, "-o", obj]
        procs.append(subprocess.Popen(cmd))
        objs.append(obj)
    assert all(p.wait() == 0 for p in procs)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", ALT], check=True)
    print("built", ALT)


def main():
    import torch
    from tracknetv3_amd import _lib, ops
    dev = torch.device("cuda", 0)

    def timeit(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    libs = {"scalar": _lib.library_path(), "imm": ALT}
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * (2.0 / (cin * 9) ** 0.5)
        mean, scale, shift = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        row, ys = {}, {}
        for rep in range(3):
            for tag, path in libs.items():
                _lib.use_library(path)
                for v in ((0, 2) if cout % 128 == 0 else (0,)):
                    u = ops.pack_wino43_weights(wt, variant=v)
                    fn = lambda: ops.conv3x3_wino43(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True, variant=v)      # noqa: E731
                    ys[(tag, v)] = fn()
                    row.setdefault(f"{tag}_geometry{'128' if (v == 0 and cout % 128 == 0) else '64'}", []).append(round(timeit(fn), 4))
        row["bit_identical"] = all(torch.equal(ys[("imm", v)], ys[("scalar", v)]) for (t, v) in ys if t == "imm")
        out[f"{cin}->{cout}@{h}x{w}"] = {k: (min(v) if isinstance(v, list) else v) for k, v in row.items()}
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(out[f"{cin}->{cout}@{h}x{w}"]), flush=True)
    _lib.use_library(libs["scalar"])
    od = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "a_imm_offsets_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
