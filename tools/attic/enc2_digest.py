"""sha256 of conv2_gdn48's output on seeded inputs (A/B of two library builds: SC2_LIB=...)."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sc2bench_amd as S
hip = S.hip
dev = torch.device('cuda:0')
for N, H, W in ((256, 110, 112), (3, 110, 112), (9, 37, 112), (4, 61, 150), (2, 257, 258)):   # (W != 112: the SEG instantiations)
    torch.manual_seed(N)
    x = torch.randn(N, H, W, 96, device=dev).to(torch.bfloat16)
    w = torch.randn(48, 96, 5, 5, device=dev) / 49
    wp = hip.pack_conv_weight(w, hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR)
    gp = hip.pack_weight_fragments(torch.nn.functional.pad(torch.rand(48, 48, device=dev) * 0.05, (0, 16)))
    beta = torch.rand(48, device=dev) + 0.5
    for inverse in (False, True):
        outs = [hip.conv2_gdn48_fwd(x, wp, gp, beta, inverse) for _ in range(3)]
        torch.cuda.synchronize()
        d = [hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16] for o in outs]
        print(N, H, W, inverse, d[0], 'repeatable' if len(set(d)) == 1 else 'NOT REPEATABLE {}'.format(d))
