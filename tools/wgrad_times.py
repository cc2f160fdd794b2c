"""Development aid: the weight-gradient launches of the stage-1 step at bs 256, per pixel-split target (policy wgrad_wgs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc2bench_amd import hip  # noqa: E402


def timeit(fn, iters=6, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


dev = torch.device('cuda:0')
N = 256
cases = [('igdn1 gamma 512 @56', 512, 512, 1, 0, 56, True), ('dec.conv4 256->256 k2 p1 @56', 256, 256, 2, 1, 56, False),
         ('dec.conv2 512->256 k2 p0 @56', 512, 256, 2, 0, 56, False), ('enc.conv2 96->48 k5 s2 @112', 96, 48, 5, 2, 112, False),
         ('igdn3 gamma 256 @55', 256, 256, 1, 0, 55, True), ('dec.conv0 24->512 k2 p1 @55', 24, 512, 2, 1, 55, False)]
if len(sys.argv) > 1 and sys.argv[1] == 'head':     # the trainable tail of stage 2 (layer2 .. layer4 at bs 256): small pixel counts, wide weights
    cases = [('l2 c1 512->128 @28', 512, 128, 1, 0, 28, False), ('l2 c2 128->128 k3 @28', 128, 128, 3, 1, 28, False),
             ('l2 c3 128->512 @28', 128, 512, 1, 0, 28, False), ('l3 c1 1024->256 @14', 1024, 256, 1, 0, 14, False),
             ('l3 c2 256->256 k3 @14', 256, 256, 3, 1, 14, False), ('l3 c3 256->1024 @14', 256, 1024, 1, 0, 14, False),
             ('l4 c1 2048->512 @7', 2048, 512, 1, 0, 7, False), ('l4 c2 512->512 k3 @7', 512, 512, 3, 1, 7, False),
             ('l4 c3 512->2048 @7', 512, 2048, 1, 0, 7, False)]
for name, cin, cout, k, pad, hw, xabs in cases:
    stride = 2 if k == 5 else 1
    oh = (hw + 2 * pad - k) // stride + 1
    x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
    g = torch.randn(N, oh, oh, cout, device=dev).to(torch.bfloat16)
    tf = 2.0 * N * oh * oh * cout * cin * k * k / 1e12
    for ct in ((0, 128) if (cout > 128 or cout <= 64) else (128,)):     # (0: the 256- / 64-channel tile where it applies; 128: always 128 x 128)
        row = []
        for wgs in (0, 64, 128, 256, 512, 768, 1024, 1536, 2048, 4096):   # (0: the launcher's own choice)
            hip.configure(wgrad_wgs=wgs, wgrad_ct=ct)
            ms = timeit(lambda: hip.conv2d_wgrad(x, g, k, k, stride, pad, x_abs=xabs))
            row.append('{}: {:.3f}'.format(wgs or 'auto', ms))
        print('{:<32}{:.3f} TFLOP  tile {:<4}'.format(name, tf, (256 if cout > 128 else 64) if ct == 0 else 128) + '  '.join(row))
hip.configure(wgrad_wgs=0, wgrad_ct=0)
