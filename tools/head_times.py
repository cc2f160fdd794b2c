"""Per-launch timing of the fused inference head (layer2..fc) at bs 256: where the 53 launches spend their time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from sc2bench_amd import hip
from tools import env_policy  # (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
dev = torch.device('cuda:0')
m = B.build_model(dev)
N = 256
x = torch.randn(N, 56, 56, 256, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
with torch.no_grad():
    for _ in range(3): m.head(x)
    torch.cuda.synchronize()
    with hip.KernelTimer() as t:
        for _ in range(5): m.head(x)
        torch.cuda.synchronize()
hd = m._hip_head
rows = []
shapes = {}
H = 56
for (c1, c2, c3, ds) in hd.blocks:
    pass
tot = 0
for tag, (cnt, ms) in sorted(t.summary().items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print('{:<16} {:7.3f} ms'.format(tag, ms))
print('total', tot)
