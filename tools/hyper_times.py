"""Development aid: stage timings of the MSHP / SHP bottleneck at the BASELINE shape (bs x 3 x 224 x 224)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import hip
ap = argparse.ArgumentParser(); ap.add_argument('--bs', type=int, default=256); ap.add_argument('--name', default='MSHPBasedResNetBottleneck')
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.get_layer(args.name).eval().to(dev)
with torch.no_grad():
    # latent of a few units, scales spread over the table (an untrained h_s would pin every scale at the 0.11 floor
    # and turn every symbol into a bypass-coded escape)
    m.g_a[4].weight.mul_(12.0); m.h_a[2].weight.mul_(12.0); m.h_s[4].weight.mul_(40.0)
    m.h_s[4].weight.abs_()
m.update()
x = torch.rand(args.bs, 3, 224, 224, device=dev)
def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, r
with torch.no_grad():
    t, y = timeit(lambda: m.analysis(x)); print('g_a            %8.3f ms' % t)
    t, z = timeit(lambda: m.hyper_analysis(y)); print('h_a            %8.3f ms' % t)
    eb, gc = m.entropy_bottleneck, m.gaussian_conditional
    t, zc = timeit(lambda: eb.compress_device(z)); print('z encode       %8.3f ms' % t)
    zbuf, zoff, znb, _ = zc
    t, zh = timeit(lambda: eb.decompress_device(zbuf, zoff, znb, tuple(z.shape[-2:]), want_f32=False, want_nhwc=True)); print('z decode       %8.3f ms' % t)
    t, p = timeit(lambda: m.hyper_synthesis(zh[1])); print('h_s            %8.3f ms' % t)
    sc, mu = m._params(p)
    t, idx = timeit(lambda: gc.build_indexes(sc)); print('build_indexes  %8.3f ms' % t)
    t, yc = timeit(lambda: gc.compress_device(y, idx, mu)); print('y encode       %8.3f ms  (%.1f B/img)' % (t, yc[2].float().mean().item()))
    t, yd = timeit(lambda: gc.decompress_device(yc[0], yc[1], yc[2], idx, mu, want_f32=False, want_nhwc=True), iters=2); print('y decode       %8.3f ms' % t)
    t, out = timeit(lambda: m.synthesis_nhwc(yd[1])); print('g_s            %8.3f ms' % t)
