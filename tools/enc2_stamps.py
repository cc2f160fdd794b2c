"""Development aid: phase timeline of the fused conv2+GDN48 kernel from a -DSC2_ENC2_STAMPS=1 build
(bash tools/build_variant.sh stamps2 -DSC2_ENC2_STAMPS=1; SC2_LIB=tools/variants/lib_stamps2.so SC2_ENC2_STAMPS=/tmp/st2.bin python tools/enc2_stamps.py)."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module('sc2-benchmark_amd')
hip = S.hip
dev = torch.device('cuda:0')
torch.manual_seed(0)
N = 256
x = torch.randn(N, 112, 112, 96, device=dev).to(torch.bfloat16)
w = torch.randn(48, 96, 5, 5, device=dev) / 49
wp = hip.pack_conv_weight(w, hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR)
gp = hip.pack_weight_fragments(torch.nn.functional.pad(torch.eye(48, device=dev) * 0.1, (0, 16)))
beta = torch.ones(48, device=dev)
path = os.environ['SC2_ENC2_STAMPS']
for _ in range(3):
    hip.conv2_gdn48_fwd(x, wp, gp, beta)
torch.cuda.synchronize()
raw = np.fromfile(path, dtype=np.uint64).astype(np.int64)
if raw.size == 8 * 4 * 16 * 40:   # -DSC2_ENC2_STAMPS=2: slots 10 cb + t inside slab cb (t = 0: first fragments issued, 1 + q: behind tap q), phases in 30 .. 39
    fine = raw.reshape(8, 4, 16, 40)
    for wg in (0, 3):
        for wave in range(4):
            t = fine[wg, wave, 2:10]
            for cb in range(3):
                top = t[:, 30 + 1 + 2 * cb]
                row = [(t[:, 10 * cb] - top).mean()] + [(t[:, 10 * cb + 1 + q] - t[:, 10 * cb + q]).mean() for q in range(7)]
                if cb == 0 and t[:, 8].any():   # -DSC2_ENC2_STAMPS=3: the groups of tap 0 of one slab (slots 8, 9, 18, 19, 28, 29)
                    gs = [8, 9, 18, 19, 28, 29]
                    print('   groups of tap 0:', ' '.join('{:.0f}'.format((t[:, b] - t[:, a]).mean()) for a, b in zip(gs[:-1], gs[1:])))
                print('wg', wg, 'wave', wave, 'slab', cb, 'first reads {:.0f} | taps '.format(row[0]) + ' '.join('{:.0f}'.format(v) for v in row[1:]),
                      '| slab {:.0f}'.format((t[:, 30 + 2 + 2 * cb] - top).mean()))
    st = fine[..., 30:]
    st = np.concatenate([st, np.zeros(st.shape[:-1] + (2,), np.int64)], -1)
else:
    st = raw.reshape(8, 4, 16, 12)   # [wg][wave][unit][stamp]
names = ['wait0+bar', 'slab0', 'wait1+bar', 'slab1', 'wait2+bar', 'slab2', 'reduce', 'gdn', 'store+bar', 'loop']
for wg in (0, 3):
    for wave in range(4):
        t = st[wg, wave]
        seg = np.diff(t[2:10, :10], axis=1)
        nxt = t[3:11, 0] - t[2:10, 9]
        print('wg', wg, 'wave', wave, ' '.join('{}={:.0f}'.format(n, v) for n, v in zip(names, list(seg.mean(0)) + [nxt.mean()])),
              'unit total', (t[3:11, 0] - t[2:10, 0]).mean())
