"""A/B in ONE process: 4-wave 128-row kernel vs 8-wave staggered 256-row kernel on the MFMA-bound layer shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sc2bench_amd import hip

dev = torch.device('cuda:0')
torch.manual_seed(0)
N = 256
shapes = [  # name, Cin, Cout, k, pad, H (input spatial), gdn
    ('dec.conv2 512->256 k2', 512, 256, 2, 0, 56, False),
    ('dec.conv4 256->256 k2p1', 256, 256, 2, 1, 55, False),
    ('igdn512', 512, 512, 1, 0, 56, True),
    ('igdn256', 256, 256, 1, 0, 55, True),
    ('head 3x3 128', 128, 128, 3, 1, 28, False),
    ('head 1x1 512->128', 512, 128, 1, 0, 28, False),
    ('head 1x1 128->512', 128, 512, 1, 0, 28, False),
    ('head 1x1 256->1024', 256, 1024, 1, 0, 14, False),
]
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, cin, cout, k, pad, H, gdn in shapes:
    x = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16)
    w = hip.pack_conv_weight(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
    beta = torch.ones(cout, device=dev)
    OH = H + 2 * pad - k + 1
    flops = 2.0 * N * OH * OH * cout * cin * k * k
    def run():
        if gdn:
            return hip.conv2d_fwd(x, w, cout, 1, 1, 1, 0, a_op=hip.AOP_ABS, epilogue=hip.EPI_IGDN, ep_x=x, ep_beta=beta)
        return hip.conv2d_fwd(x, w, cout, k, k, 1, pad)
    res = {}
    for rnd in range(3):
        for mode in ('small', 'big'):
            os.environ.pop('SC2_CONV_NO_BIG', None); os.environ.pop('SC2_CONV_FORCE_BIG', None)
            os.environ['SC2_CONV_NO_BIG' if mode == 'small' else 'SC2_CONV_FORCE_BIG'] = '1'
            res.setdefault(mode, []).append(timeit(run))
    s, b = min(res['small']), min(res['big'])
    print('{:<28} small {:7.3f} ms {:7.1f} TF | big {:7.3f} ms {:7.1f} TF'.format(name, s, flops / s / 1e9, b, flops / b / 1e9))
