"""Fixed cost per launch of the head's layer kinds: one representative launch per kind (tools/k_times.head_layer_rows) at bs 64 .. 512,
t(bs) = t0 + k bs by least squares."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.k_times import head_layer_rows, timeit
res = {}
for bs in (64, 128, 256, 512):
    with torch.no_grad():
        for name, fn, flops in head_layer_rows(bs):
            res.setdefault(name, []).append((bs, timeit(fn, 40)))
    torch.cuda.empty_cache()
print('{:<34}'.format('layer') + ''.join('{:>10}'.format('bs %d' % b) for b in (64, 128, 256, 512)) + '{:>12}{:>14}'.format('t0 (us)', 'us per 256'))
for name, pts in res.items():
    b = np.array([p[0] for p in pts], float); t = np.array([p[1] for p in pts], float) * 1e3
    k, t0 = np.polyfit(b, t, 1)
    print('{:<34}'.format(name) + ''.join('{:>10.1f}'.format(v) for v in t) + '{:>12.1f}{:>14.1f}'.format(t0, k * 256))
