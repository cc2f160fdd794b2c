"""Development aid: the f32 conv kernel against torch CPU on a few shapes, reporting WHICH 16-pixel row tiles differ (found the
asm wait without a data dependence in round 4: only the first row tile of some waves was wrong).   python tools/attic/dbg_f32.py"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sc2bench_amd as S
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
hip = S.hip
dev = torch.device('cuda:0')
for (cin, cout, k, s, p, hw) in [(3, 96, 5, 2, 2, (37, 50)), (96, 48, 5, 2, 2, (28, 31)), (4, 5, 1, 1, 0, (7, 5)), (16, 16, 1, 1, 0, (8, 16))]:
    g = torch.Generator().manual_seed(cin * 131 + cout)
    x = torch.randn(3, cin, hw[0], hw[1], generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x, w, None, s, p)
    xh = hip.nchw_f32_to_nhwc_f32(x.to(dev))
    wf = hip.pack_conv_f32(w.to(dev))
    y = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p).permute(0, 3, 1, 2).cpu()
    err = (y - ref).abs()
    bad = err > 1e-4
    print((cin, cout, k), 'bad fraction', bad.float().mean().item(), 'max', err.max().item())
    if bad.any():
        idx = bad.nonzero()
        print('  bad n', sorted(set(idx[:, 0].tolist())), 'c', sorted(set(idx[:, 1].tolist()))[:20], 'h', sorted(set(idx[:, 2].tolist()))[:40])
        M = ref.shape[0] * ref.shape[2] * ref.shape[3]
        lin = (idx[:, 0] * ref.shape[2] + idx[:, 2]) * ref.shape[3] + idx[:, 3]
        tiles = sorted(set((lin // 16).tolist()))
        print('  M', M, 'bad 16-px tiles', tiles[:40], '... of', (M + 15) // 16)
        # per k-step: recompute partial sums to find which steps are wrong? report err of one bad element
        n, c, h, ww = idx[0].tolist()
        print('  first bad', idx[0].tolist(), 'got', y[n, c, h, ww].item(), 'ref', ref[n, c, h, ww].item())
