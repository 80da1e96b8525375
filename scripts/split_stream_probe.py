"""Does splitting ONE inference batch over two HIP streams fill the tail bubbles of the per-layer launches?  Times the
TrackNet(27, 8) eval forward on a batch of 10: whole batch on one stream vs halves (5 + 5), (6 + 4) and thirds on separate
streams, same outputs.  usage: split_stream_probe.py [batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd.model import TrackNet
from tracknetv3_amd.utils import synth


def main():
    dev = torch.device("cuda", 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    model = synth.init_state_(TrackNet(27, 8), 31, calibrated=True).to(dev).eval()
    x = torch.rand(n, 27, 288, 512, device=dev)
    main_s = torch.cuda.current_stream(dev)
    sides = [torch.cuda.Stream(dev) for _ in range(3)]

    def whole():
        return model(x)

    def split(parts):
        def run():
            ev = torch.cuda.Event()
            ev.record(main_s)
            outs, lo = [], 0
            for i, p in enumerate(parts):
                xi = x[lo:lo + p]
                lo += p
                if i == 0:
                    outs.append(model(xi))
                else:
                    s = sides[i - 1]
                    s.wait_event(ev)
                    with torch.cuda.stream(s):
                        outs.append(model(xi))
            for i in range(1, len(parts)):
                main_s.wait_stream(sides[i - 1])
            return torch.cat(outs, 0)
        return run

    def timeit(fn, reps=10):
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, y

    out = {}
    with torch.no_grad():
        ms, ref = timeit(whole)
        out["whole"] = round(ms, 4)
        for name, parts in (("5+5", (n // 2, n - n // 2)), ("6+4", (n * 6 // 10, n - n * 6 // 10)), ("4+3+3", (n - 2 * (n * 3 // 10), n * 3 // 10, n * 3 // 10)),
                            ("7+3", (n * 7 // 10, n - n * 7 // 10))):
            ms, y = timeit(split(parts))
            out[name] = {"ms": round(ms, 4), "equal": bool(torch.equal(y, ref))}
        ms, _ = timeit(whole)
        out["whole_again"] = round(ms, 4)
    print(json.dumps(out))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "split_stream_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
