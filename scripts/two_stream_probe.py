"""Does running independent batches on two HIP streams hide the batch-10 tail of the conv launches?
Times K TrackNet forwards (batch 10) issued round-robin on 1, 2 and 3 streams."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd.utils.general import get_model


def main():
    dev = torch.device("cuda", 0)
    m = get_model("TrackNet", 8, "concat").to(dev).eval()
    out = {}
    with torch.no_grad():
        for ns in (1, 2, 3, 1, 2):
            streams = [torch.cuda.Stream(dev) for _ in range(ns)]
            xs = [torch.rand(10, 27, 288, 512, device=dev) for _ in range(ns)]
            for k in range(2 * ns):
                with torch.cuda.stream(streams[k % ns]):
                    m(xs[k % ns])
            torch.cuda.synchronize(dev)
            K = 24
            t0 = time.perf_counter()
            for k in range(K):
                with torch.cuda.stream(streams[k % ns]):
                    m(xs[k % ns])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            out.setdefault(f"streams_{ns}", []).append({"ms_per_step": round(dt / K * 1e3, 3), "frames_per_s": round(80 * K / dt, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
