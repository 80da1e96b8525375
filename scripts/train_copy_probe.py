"""K training steps of the configs[2] shard (FusedAdam, device-side mixup draws) for `rocprofv3 --memory-copy-trace`: run it with
two different K -- the number of memory copies must not depend on K (a step issues no H2D / D2H copy and no host sync).
usage: train_copy_probe.py K [torch]      ("torch": torch.optim.Adam + host-side mixup draws, the round-1 protocol, for contrast)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd.optim import FusedAdam
from tracknetv3_amd.parallel import TrackNetTrainer
from tracknetv3_amd.utils import synth
from tracknetv3_amd.utils.general import get_model

k = int(sys.argv[1])
legacy = len(sys.argv) > 2 and sys.argv[2] == "torch"
dev = torch.device("cuda:0")
model = synth.init_state_(get_model("TrackNet", 8, "concat"), 31, calibrated=False).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3) if legacy else FusedAdam(model.parameters(), lr=1e-3)
tr = TrackNetTrainer(model, opt, alpha=0.5, seed=13, device_rng=not legacy)
x = torch.rand((10, 27, 288, 512), device=dev)
y = synth.disc_heatmaps(10, 8, 288, 512, 77, device=dev)
torch.cuda.synchronize()
for _ in range(k):
    loss = tr.step(x, y)
torch.cuda.synchronize()
print("steps", k, "loss", float(loss))
