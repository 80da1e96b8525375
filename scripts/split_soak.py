"""Soak of the split-stream eval forward: 300 forwards of a batch of 10 (allocated / reserved MiB every 50), then changing batch
sizes; allocated memory must stay flat (side-stream allocations are recycled by the caching allocator's per-stream pools)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tracknetv3_amd.model import TrackNet
from tracknetv3_amd.utils import synth
dev = torch.device("cuda", 0)
m = synth.init_state_(TrackNet(27, 8), 31, calibrated=True).to(dev).eval()
x = torch.rand(10, 27, 288, 512, device=dev)
res = []
for i in range(300):
    y = m(x)
    if i % 50 == 0:
        torch.cuda.synchronize()
        res.append((i, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
torch.cuda.synchronize()
print(res)
# different batch sizes interleaved
for n in (10, 7, 4, 16, 10, 5):
    y = m(torch.rand(n, 27, 288, 512, device=dev))
torch.cuda.synchronize()
print("final", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
