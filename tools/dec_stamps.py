"""Development aid: phase timeline of the fused conv2x2+GDN512 kernel from a -DSC2_DEC_STAMPS=1 build
(SC2_LIB=tools/variants/lib_stamps.so SC2_DEC_STAMPS=/tmp/st.bin python tools/dec_stamps.py)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import hip
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.FPBasedResNetBottleneck().eval().to(dev)
d0, h1 = m.decoder[0], m.decoder[1]
yh = torch.randn(256, 55, 55, 24, device=dev).to(torch.bfloat16)
beta, gamma = h1.effective_fragments()
path = os.environ['SC2_DEC_STAMPS']
with torch.no_grad():
    for _ in range(3):
        hip.conv2x2_gdn512_fwd(yh, d0.packed_weight(hip.K_TAP_MAJOR), gamma, beta, True)
    torch.cuda.synchronize()
st = np.fromfile(path, dtype=np.uint64).reshape(8, 8, 16, 8).astype(np.int64)   # [wg][wave][tile][stamp]
names = ['phase1+imgwrite', 'barrier1', 'phase2', 'patch st+epilogue', 'load_w+barrier2', 'readout', 'barrier3', 'loop']
for wg in (0, 3):
    for wave in (0, 4, 7):
        t = st[wg, wave]
        ok = t[:, 7] > 0
        seg = np.diff(t[ok][2:10], axis=1)                       # skip the first two tiles
        nxt = t[ok][3:11, 0] - t[ok][2:10, 7]
        print('wg', wg, 'wave', wave, 'mean cycles per segment:',
              ' '.join('{}={:.0f}'.format(n, v) for n, v in zip(names, list(seg.mean(0)) + [nxt.mean()])),
              'tile total', (t[ok][3:11, 0] - t[ok][2:10, 0]).mean())
