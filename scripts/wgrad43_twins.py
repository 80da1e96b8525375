"""Timing twins and candidate schedules of the F(4x4) weight-gradient kernel (kernels/wgrad_wino43_mfma.h, WgradWino43Sw) on TrackNet's
plain-layer shapes at batch 10: ms per call (kernel + fold) and, for the Timeline twins, s_memtime cycles per step of one mid-grid workgroup
split into "issuing" (step start -> the end-of-step waits) and "waiting" (the waits + the barrier), per wave group.
Switch bits: 1 no X DMA, 2 no dY loads, 4 no Yh transform, 8 no V transform, 16 no MFMAs, 128 no operand reads (WRONG results);
32 dY loads behind quad 3, 512 column passes behind quads 0-1 and dY loads behind quad 1, 256 three raw stages (results bit-identical).
  PARTS=custom CUSTOM_CMD="python scripts/wgrad43_twins.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tracknetv3_amd import _lib, ops
import diaglib

SHAPES = ((64, 64, 288, 512), (128, 128, 144, 256), (256, 256, 72, 128), (512, 512, 36, 64), (27, 64, 288, 512), (64, 128, 144, 256))
# 1000 + switches: round 5's schedule of the kernel (kernels/wgrad_wino43_r5_mfma.h; 1000 and 1064 are instantiated) -- the same-session A/B reference
SWITCHES = [int(v) for v in os.environ.get("W43_SW", "1000 0 32 512 256 288 768 1 2 3 4 8 12 16 128 15 143").split()]
TIMELINES = [int(v) for v in os.environ.get("W43_TL", "1064 64 576 832").split()]


def call(x, dz, dw, ws, sw):
    n, cout, h, w = (int(v) for v in dz.shape)
    diaglib.check(diaglib.load().tnv3_diag_conv3x3_wgrad_wino(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(ws), ws.numel() * 8, n, int(x.shape[1]), cout, h, w,
                                                               (9000 + sw - 1000 if sw >= 1000 else 8000 + sw), _lib.stream_ptr(dz)))


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        dz = torch.randn(10, cout, h, w, device=dev) * 0.1
        ref = ops.conv3x3_wgrad_wino(x, dz, variant=8)
        nbytes = _lib.load().tnv3_conv3x3_wgrad_wino_workspace_bytes(10, cin, cout, h, w)
        ws = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        dw = torch.empty_like(ref)
        row = {"product_ms": round(timeit(lambda: ops.conv3x3_wgrad_wino(x, dz, variant=8)), 4)}
        for rep in range(int(os.environ.get("W43_REPS", "2"))):      # (interleaved repetitions; the last one's time is kept, all of them are listed)
            for sw in SWITCHES:
                call(x, dz, dw, ws, sw)
                d = {"ms": round(timeit(lambda: call(x, dz, dw, ws, sw)), 4)}
                if not ((sw % 1000) & (1 | 2 | 4 | 8 | 16 | 128)):
                    d["bit_identical_to_product"] = bool(torch.equal(dw, ref))
                d["all_ms"] = row.get(f"sw{sw}", {}).get("all_ms", []) + [d["ms"]]
                row[f"sw{sw}"] = d
        for sw in TIMELINES:
            ws[:128].zero_()
            call(x, dz, dw, ws, sw)
            torch.cuda.synchronize()
            tl = ws[:64].cpu().reshape(8, 8).double()
            steps = max(float(tl[0, 2]), 1.0)
            row[f"tl{sw}"] = {"steps": steps, "cycles_per_step": round(float(tl[:, 3].mean()) / steps, 1),
                              "yh_waves_issuing_waiting": [round(float(tl[:4, 0].mean()) / steps, 1), round(float(tl[:4, 1].mean()) / steps, 1)],
                              "v_waves_issuing_waiting": [round(float(tl[4:, 0].mean()) / steps, 1), round(float(tl[4:, 1].mean()) / steps, 1)],
                              "bit_identical_to_product": bool(torch.equal(dw, ref))}
        out[f"{cin}->{cout}@{h}x{w}"] = row
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(row), flush=True)
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "wgrad43_twins.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
