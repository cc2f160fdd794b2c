"""Host range coder throughput on this machine: sc2_rans_code_host (encode + decode) of 256 streams x 72 600 symbols by thread count,
with fresh and with reused output buffers.   python tools/host_coder_scaling.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc2bench_amd import hip
g = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'fp_golden.pt'), weights_only=False)
cdf, ln, off = g['quantized_cdf'].numpy(), g['cdf_length'].numpy().reshape(-1), g['offset'].numpy().reshape(-1)
t = hip.HostRansTables(cdf, ln, off)
rng = np.random.RandomState(0)
hw = 55 * 55
one = np.concatenate([np.clip(np.round(rng.randn(hw) * 1.5).astype(np.int32), off[c], off[c] + ln[c] - 3) for c in range(24)]).astype(np.int32)
sym = np.ascontiguousarray(np.tile(one, (256, 1)))
dec = np.empty_like(sym)
print('cores', hip.host_cores())
for thr in (8, 16, 32, 64, 128, 256):
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        buf, o, nb, st = hip.rans_code_host(t, sym, hw, dec, threads=thr)
        ts.append(1e3 * (time.perf_counter() - t0))
    assert np.array_equal(dec, sym) and not st.any()
    scratch, tr = {}, []
    for rep in range(5):
        t0 = time.perf_counter()
        buf, o, nb, st = hip.rans_code_host(t, sym, hw, dec, threads=thr, scratch=scratch)
        tr.append(1e3 * (time.perf_counter() - t0))
    print('threads {:3d}: fresh rows {} ms | reused rows {} ms'.format(thr, ' '.join('{:.2f}'.format(v) for v in ts), ' '.join('{:.2f}'.format(v) for v in tr)))
