"""Debug driver (round 4): the window-plane head kernels at the bs-256 shapes of the head, one launch at a time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sc2bench_amd as S
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
hip = S.hip
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(1)
for cin, cout, HW in ((128, 128, 28), (256, 256, 14), (512, 512, 7)):
    print('conv3x3_win', cin, cout, HW, flush=True)
    x = torch.randn(N, HW, HW, cin, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    out = hip.conv3x3_win_fwd(x, hip.pack_conv3x3_win(w), b, relu=True)
    torch.cuda.synchronize()
    ref = hip.conv2d_fwd(x, hip.pack_conv_weight(w), cout, 3, 3, 1, 1, epilogue=hip.EPI_BIAS_RELU, ep_beta=b)
    torch.cuda.synchronize()
    print('   max diff vs tile kernel', (out.float() - ref.float()).abs().max().item(), flush=True)
for cin, cout, HW, stride, res in ((512, 128, 28, 1, False), (2048, 512, 7, 1, False), (512, 2048, 7, 1, True), (512, 1024, 28, 2, False)):
    print('conv1x1_win', cin, cout, HW, stride, res, flush=True)
    x = torch.randn(N, HW, HW, cin, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    OH = (HW - 1) // stride + 1
    r = torch.randn(N, OH, OH, cout, generator=g).to(dev).to(torch.bfloat16) if res else None
    out = hip.conv1x1_win_fwd(x, hip.pack_conv_win(w), b, stride=stride, relu=True, residual=r)
    torch.cuda.synchronize()
    print('   ok', tuple(out.shape), float(out.float().abs().mean()), flush=True)
print('done')
