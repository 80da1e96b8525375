#!/usr/bin/env python
"""Attribute the conv kernel's matrix-pipe idle time: production vs 'no re-staging' vs 'no re-staging, no barriers'."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tracknetv3_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import diaglib

def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out = {}
    for (name, cin, cout, h, w) in (("64to64_288x512", 64, 64, 288, 512), ("192to64_288x512", 192, 64, 288, 512),
                                    ("128to128_144x256", 128, 128, 144, 256), ("512to512_36x64", 512, 512, 36, 64)):
        n = 10
        x = torch.rand(n, cin, h, w, device=dev)
        wp = ops.pack_conv3x3_weights(torch.empty(cout, cin, 3, 3, device=dev).uniform_(-0.05, 0.05))
        y = torch.empty(n, cout, h, w, device=dev)
        fl = 2.0 * 9 * cin * cout * h * w * n
        row = {}
        for cfg in (12, 10, 11):
            info = ops.conv3x3_config_info(cfg)
            if cout % info["m_block"]:
                continue
            for diag in (0, 1, 2):
                f = lambda: diaglib.conv3x3_forward(x, wp, y, cfg, diag)
                for _ in range(2): f()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): f()
                e1.record(); torch.cuda.synchronize(dev)
                row[f"cfg{cfg}_diag{diag}"] = round(fl / (e0.elapsed_time(e1) / 5) / 1e9, 1)
        out[name] = row
        print(name, row, flush=True)
    print(json.dumps(out))

if __name__ == "__main__":
    main()
