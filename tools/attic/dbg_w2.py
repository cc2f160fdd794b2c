"""Debug driver of the window-plane decoder kernel: repeated launches against the tile kernels, where the outputs differ.
(SC2_W2_DBG=1 in the environment: the library under test was built with -DSC2_W2_DBG_OUT=1 -- bash tools/build_variant.sh dbgx
-DSC2_W2_DBG_OUT=1, SC2_LIB=tools/variants/lib_dbgx.so -- and stores the conv output instead of the GDN1 output.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['SC2_W2_RUN'] = sys.argv[1] if len(sys.argv) > 1 else '5'
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 512
import sc2bench_amd as S
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
dev = torch.device('cuda:0')
pad, N, H = 0, 3, 56
W = 56
tot = 0
for trial in range(3):
    g = torch.Generator().manual_seed(cin + N + H + trial)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(256, cin, 2, 2, generator=g) / (4 * cin) ** 0.5
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    order = S.hip.preferred_k_order(cin, 2, 2)
    wp = S.hip.pack_conv_weight(w.to(dev), order)
    gdn = S.GDN1(256, inverse=True).to(dev)
    with torch.no_grad():
        gdn.gamma.add_(0.02 * torch.rand(256, 256, generator=g).to(dev))
    conv = S.hip.conv2d_fwd(x_nhwc, wp, 256, 2, 2, 1, pad, k_order=order)
    ref = gdn.forward_nhwc(conv) if fused else conv
    if os.environ.get('SC2_W2_DBG') == '1':
        ref = conv
    wf = S.hip.pack_conv2x2_win(w.to(dev), gdn.gamma_reparam(gdn.gamma).detach() if fused else None)
    beta = gdn.beta_reparam(gdn.beta).detach().float().contiguous() if fused else None
    for rep in range(4):
        out = S.hip.conv2x2_win_fwd(x_nhwc, wf, pad, beta=beta, inverse=True)
        d = (out != ref).nonzero()
        tot += d.shape[0]
        if d.shape[0]:
            print('trial', trial, 'rep', rep, 'diff', d.shape[0], 'tiles', sorted(set((d[:, 0] * 14 + d[:, 1] // 4).tolist())),
                  'chans', sorted(set((d[:, 3] // 32).tolist())), 'e', sorted(set((d[:, 3] % 8).tolist())),
                  'rows%4', sorted(set((d[:, 1] % 4).tolist())), 'cols', sorted(set(d[:, 2].tolist()))[:20])
print('run', os.environ['SC2_W2_RUN'], 'fused', fused, 'cin', cin, 'total differing elements', tot)
