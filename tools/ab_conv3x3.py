"""Stand-alone time of the head's 3x3 launches (one per map size, bs 256) -- the rows an A/B of two library builds needs
(SC2_LIB=tools/variants/lib_<name>.so selects the build):   python tools/ab_conv3x3.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from tools.k_times import timeit
dev = torch.device('cuda:0')
m = B.build_model(dev)
hd = m._hip_head_for_eval()
rows = []
with torch.no_grad():
    for (c1, c2, c3, ds), (hw, cin) in zip([hd.blocks[1], hd.blocks[5], hd.blocks[11], hd.blocks[0], hd.blocks[4]],
                                           [((28, 28), 128), ((14, 14), 256), ((7, 7), 512), ((56, 56), 128), ((28, 28), 256)]):
        x = torch.randn(256, hw[0], hw[1], cin, device=dev).to(torch.bfloat16)
        from sc2bench_amd import hip
        rows.append((c2.tag, timeit(lambda: c2(x, hip.EPI_BIAS_RELU), 50, 5)))
print(' '.join('{} {:.4f}'.format(t, ms) for t, ms in rows), ' sum {:.4f} ms'.format(sum(ms for _, ms in rows)))
