"""Is the encoder faster in slices whose intermediate tensor fits the 256 MB memory-side cache?  analysis() of 256 images as 1, 2, 4, 8 launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sc2bench_amd as S
from tools.k_times import timeit
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.FPBasedResNetBottleneck().eval().to(dev)
x = torch.rand(256, 3, 224, 224, device=dev)
with torch.no_grad():
    for parts in (1, 2, 4, 8, 1, 4):
        n = 256 // parts
        xs = [x[i * n:(i + 1) * n] for i in range(parts)]
        def run():
            for xi in xs:
                m.analysis(xi)
        print('analysis(256 images) as {} launches of {}: {:.4f} ms'.format(parts, n, timeit(run, 30)))
