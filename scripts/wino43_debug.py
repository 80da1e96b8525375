import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.nn.functional as F
from tracknetv3_amd import ops
from test_emu_kernels import T
d = torch.device("cuda", 0)
for case in [(1,16,64,8,64),(2,20,128,16,64),(1,24,64,8,64),(1,16,128,8,64),(2,16,64,8,64),(1,16,64,16,64),(2,64,64,288,512)]:
    n,cin,cout,h,w = case
    x, wt = torch.relu(T((n,cin,h,w),491)).to(d), T((cout,cin,3,3),492,-0.3,0.3).to(d)
    ref = F.conv2d(x.double(), wt.double(), padding=1); mag = ref.abs().max().item()
    u = ops.pack_wino43_weights(wt)
    errs = []
    for _ in range(3):
        y = ops.conv3x3_wino43(x, u, cout)
        errs.append(round((y.double()-ref).abs().max().item()/mag, 8))
    print(case, errs, flush=True)
