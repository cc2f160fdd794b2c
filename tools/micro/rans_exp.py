"""Experiment driver: time rans encode/decode from an alternative build of rans.hip (LIB env)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sc2bench_amd as S
from sc2bench_amd import hip
if os.environ.get('SC2_LIB'):
    hip.LIB_PATH = os.environ['SC2_LIB']; hip._lib = None
dev = torch.device('cuda:0')
torch.manual_seed(0)
N, C, HW = 256, 24, 3025
m = S.FPBasedResNetBottleneck().to(dev)
from oracle import cpu_ref as R
R.perturb_quantiles(m.entropy_bottleneck)
m.update()
eb = m.entropy_bottleneck
cdf, cl, off = eb._tables()
sym = torch.randint(-3, 4, (N, C * HW), dtype=torch.int32, device=dev)
def t(fn, it=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
buf, o, nb, st = hip.rans_encode_batch(sym, cdf, cl, off, index_div=HW)
print('encode ms', t(lambda: hip.rans_encode_batch(sym, cdf, cl, off, index_div=HW)), 'bytes', nb.float().mean().item())
print('decode ms', t(lambda: hip.rans_decode_batch(buf, o, nb, C * HW, cdf, cl, off, index_div=HW)))
