"""Samples GPU clock / power (rocm-smi) while (a) the register-only MFMA probe, (b) TrackNet inference and
(c) the training step run back-to-back, to tell a pipeline limit from a power/clock limit."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import diaglib


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            rec = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "speed" in kl:
                    rec["sclk"] = v
                if "mclk" in kl and "speed" in kl:
                    rec["mclk"] = v
                if "power" in kl and "w" in kl:
                    rec["power"] = v
            out.append(rec)
        except Exception as e:  # noqa: BLE001
            out.append({"err": str(e)[:80]})
        time.sleep(0.05)


def run_phase(name, fn, seconds, res):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    fn(); torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    res[name] = {"iters": n, "s": round(dt, 3), "samples": out[1:-1][:40]}


def main():
    from tracknetv3_amd import _lib, ops
    from tracknetv3_amd.utils.general import get_model
    dev = torch.device("cuda", 0)
    res = {}
    lib = _lib.load()
    buf = torch.empty(256 * 8 * 256, dtype=torch.float32, device=dev)
    run_phase("mfma_probe", lambda: diaglib.mfma_f32_probe(buf, 2048, 20000), 4.0, res)
    m = get_model("TrackNet", 8, "concat").to(dev).eval()
    x = torch.rand(10, 27, 288, 512, device=dev)
    with torch.no_grad():
        run_phase("tracknet_infer", lambda: m(x), 4.0, res)
    run_phase("idle", lambda: time.sleep(0.2), 1.0, res)
    for flag in ("--showmaxpower", "--showclocks"):
        try:
            res["rocm_smi " + flag] = json.loads(subprocess.run(["rocm-smi", flag, "--json"], capture_output=True, text=True, timeout=10).stdout)
        except Exception as e:  # noqa: BLE001
            res["rocm_smi " + flag] = str(e)[:100]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
