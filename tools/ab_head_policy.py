"""Alternating A/B of ONE host-policy switch on the wall time of the task head (layer2..fc, bs 256) and of the whole back stage
(dequantise + decoder + head), HIP events around 10 calls each, five rounds:   python tools/ab_head_policy.py head_ds_side_stream"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from sc2bench_amd import hip
field = sys.argv[1]
dev = torch.device('cuda:0')
m = B.build_model(dev)
x = B.synthetic_batch(256, dev)
feat = torch.randn(256, 56, 56, 256, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)


def wall(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    sym, hw = m.stage_front(x)
    dec, _, _ = m.stage_coder(sym, hw, dequantized=True)
    ref = None
    for rnd in range(5):
        row = []
        for v in (True, False):
            hip.configure(**{field: v})
            out = m.stage_back(dec, hw)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), 'the switch changes the result'
            row.append((wall(lambda: m.head(feat)), wall(lambda: m.stage_back(dec, hw))))
        print('round {}: head on {:.3f} off {:.3f} ms | back stage on {:.3f} off {:.3f} ms'.format(rnd, row[0][0], row[1][0], row[0][1], row[1][1]))
