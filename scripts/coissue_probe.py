"""Cost table: instruction classes next to a stream of v_mfma_f32_32x32x2_f32 on the same SIMD (libtnv3_diag.so,
kernels/coissue_probe.h).  Prints cycles per step (one step = one MFMA, or one filler group) for waves 0-3 and 4-7."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import diaglib

ROLES = {0: "idle", 1: "mfma", 2: "valu x4", 3: "ds_read_b32 x2", 4: "ds_read_b128", 5: "ds_write_b64", 6: "buffer_load_lds 16B",
         7: "mfma + 2 valu", 8: "mfma + ds_read_b128 + ds_write_b64 + 2 valu", 9: "mfma + 2 ds_read_b32"}


def main():
    dev = torch.device("cuda", 0)
    blocks, iters = 256, 64
    src = torch.rand(1 << 20, device=dev)
    out = torch.zeros(blocks * 8, dtype=torch.int64, device=dev)
    res = {}
    pairs = [(1, 0), (0, 1), (1, 1), (9, 0), (9, 9), (7, 0), (7, 7), (8, 0), (8, 8)]
    for filler in (2, 3, 4, 5, 6):
        pairs += [(filler, 0), (1, filler), (filler, 1)]
    for ra, rb in pairs:
        for _ in range(2):
            diaglib.coissue_probe(out, src, blocks, ra, rb, iters)
        torch.cuda.synchronize()
        t = out.view(blocks, 8).double().cpu()
        steps = iters * 8
        a, b = float(t[:, :4].mean()) / steps, float(t[:, 4:].mean()) / steps
        key = f"waves0-3: {ROLES[ra]:45s} | waves4-7: {ROLES[rb]}"
        res[key] = {"cycles_per_step_waves0_3": round(a, 1), "cycles_per_step_waves4_7": round(b, 1)}
        print(f"{key:110s} -> {a:7.1f} {b:7.1f}", flush=True)
    json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "coissue_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
