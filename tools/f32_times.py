"""Per-launch times of the reference-precision (f32 operand) encoder at bs 256, 224 x 224 (HIP events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sc2bench_amd import hip  # noqa: E402

dev = torch.device('cuda:0')
model = bench.build_model(dev, encoder_precision='f32')
x = bench.synthetic_batch(256, dev)
GF = {'enc.conv0.f32+enc.gdn1.f32': 411.8, 'enc.conv2.f32+enc.gdn3.f32': 737.0, 'enc.conv0.f32': 180.6, 'enc.gdn1.f32': 231.2, 'enc.conv2.f32': 722.5, 'enc.gdn3.f32': 14.5, 'enc.conv4.f32': 27.9}
with torch.no_grad():
    for _ in range(3):
        model.stage_front(x)
    torch.cuda.synchronize()
    with hip.KernelTimer() as kt:
        for _ in range(5):
            model.stage_front(x)
        torch.cuda.synchronize()
tot = 0.0
for k, (n, ms) in sorted(kt.summary().items()):
    tf = GF.get(k, 0) * 1e6 * 256 / (ms * 1e-3) / 1e12 if k in GF else 0.0
    tot += ms
    print('{:<18} {:7.3f} ms  {:6.1f} TFLOP/s  ({:.2f} of the 157 TFLOP/s f32 matrix peak)'.format(k, ms, tf, tf / 157.3))
print('total {:.3f} ms'.format(tot))
