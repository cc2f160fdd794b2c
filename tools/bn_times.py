"""Development aid: sc2_bn_train_fwd / _bwd per layer shape of the stage-2 student tail at bs 256, against torch's bn + relu."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S  # noqa: E402


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (hw, C) in ((28, 512), (28, 128), (14, 1024), (14, 256), (7, 2048), (7, 512)):
    x = torch.randn(256, hw, hw, C, device='cuda').to(torch.bfloat16)
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    y, m, r = S.hip.bn_train_fwd(x, g, b, rm, rv, 0.1, 1e-5, True)
    mb = x.numel() * 2 / 1e6
    tf = t(lambda: S.hip.bn_train_fwd(x, g, b, rm, rv, 0.1, 1e-5, True))
    tb = t(lambda: S.hip.bn_train_bwd(x, x, y, g, m, r))
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    xc = x.permute(0, 3, 1, 2).detach().requires_grad_(True)
    tt = t(lambda: torch.relu(bn(xc)))
    out = torch.relu(bn(xc))
    go = torch.ones_like(out)
    ttb = t(lambda: torch.autograd.grad(out, xc, go, retain_graph=True))
    print('hw %2d C %4d %4.0f MB  hip fwd %.3f ms (%.2f TB/s over 3 passes)  bwd %.3f ms (%.2f TB/s over 5)   torch bn+relu fwd %.3f bwd %.3f' % (
        hw, C, mb, tf, 3 * mb / tf / 1e3, tb, 5 * mb / tb / 1e3, tt, ttb))
