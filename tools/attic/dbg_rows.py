"""Development aid: where gdn512_rows' backward differs from the f32 reference (one tile, no zeros), by tile coordinates."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sc2bench_amd import hip  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(1)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
inverse = True
x = torch.randn(M, C).to(torch.bfloat16).float()
gy = torch.randn(M, C).to(torch.bfloat16).float()
gamma = (0.3 * torch.rand(C, C) / C ** 0.5 + 0.1 * torch.eye(C)).to(torch.bfloat16).float()
beta = 1.0 + 0.1 * torch.rand(C)
n = x.abs() @ gamma.t() + beta
dn = gy * x
dd = gy * n
t = dn.to(torch.bfloat16).float() @ gamma
dx = dd + torch.sign(x) * t
xd, gd = x.to(torch.bfloat16).to(dev), gy.to(torch.bfloat16).to(dev)
gf, gtf = hip.pack_weight_fragments(gamma.to(dev)), hip.pack_weight_fragments(gamma.t().contiguous().to(dev))
d_norm, dxg = hip.gdn1_rows_bwd(xd, gd, gf, gtf, beta.to(dev), inverse)
torch.cuda.synchronize()
d_norm, dxg = d_norm.float().cpu(), dxg.float().cpu()


def report(name, got, ref):
    err = (got - ref).abs() > 0.02 * ref.abs() + 0.02
    print(name, 'bad', int(err.sum()), 'of', err.numel())
    if err.any():
        rows, cols = err.nonzero(as_tuple=True)
        print('  bad rows  (count by row %% 16):', torch.bincount(rows % 16, minlength=16).tolist())
        print('  bad rows  (count by row // 16 %% 8):', torch.bincount((rows // 16) % 8, minlength=8).tolist())
        print('  bad cols  (count by wave = c // 64):', torch.bincount(cols // 64, minlength=8).tolist())
        print('  bad cols  (count by j = c %% 64 // 16):', torch.bincount((cols % 64) // 16, minlength=4).tolist())
        print('  bad cols  (count by c %% 16):', torch.bincount(cols % 16, minlength=16).tolist())
        r, c = int(rows[0]), int(cols[0])
        print('  first bad', r, c, 'got', got[r, c].item(), 'ref', ref[r, c].item(), 'dd', dd[r, c].item(), 't', t[r, c].item(), 'dn', dn[r, c].item(),
              'gy', gy[r, c].item(), 'x', x[r, c].item())


report('d_norm', d_norm, dn)
report('dx', dxg, dx)
report('dx vs dd only', dxg, dd)
report('dx vs sign*t only', dxg, torch.sign(x) * t)
print(dxg[0, :8], dx[0, :8])
