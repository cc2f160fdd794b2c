"""Bottleneck forward at the shapes of BASELINE configs 4 and 5: 513 x 513, N = 16 (VOC batch) and
800 x 1216, N = 6 (typical R-CNN batch), beside 224 x 224, N = 256.  HIP events on the current stream; algorithmic
FLOPs = 2 * MACs of the ten transforms at that shape.  Prints one JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S  # noqa: E402
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import hip  # noqa: E402


def gflop_per_image(H, W):
    def o(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h1, w1 = o(H, 5, 2, 2), o(W, 5, 2, 2)
    h2, w2 = o(h1, 5, 2, 2), o(w1, 5, 2, 2)
    h3, w3 = h2 - 1, w2 - 1
    h4, w4 = h3 + 1, w3 + 1
    h5, w5 = h4 - 1, w4 - 1
    h6, w6 = h5 + 1, w5 + 1
    macs = (h1 * w1 * 96 * 75 + h1 * w1 * 96 * 96 + h2 * w2 * 48 * 2400 + h2 * w2 * 48 * 48 + h3 * w3 * 24 * 192 +
            h4 * w4 * 512 * 96 + h4 * w4 * 512 * 512 + h5 * w5 * 256 * 2048 + h5 * w5 * 256 * 256 + h6 * w6 * 256 * 1024)
    return 2e-9 * macs


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = S.FPBasedResNetBottleneck().eval().to(dev)
    from oracle import cpu_ref as R
    R.perturb_quantiles(m.entropy_bottleneck)
    m.update()
    m.output_format = 'bf16_nhwc'
    for N, H, W in ((256, 224, 224), (16, 513, 513), (6, 800, 1216)):
        x = torch.rand(N, 3, H, W, device=dev)
        with torch.no_grad():
            latent = m.analysis(x)
            _, y_nhwc = hip.eb_dequantize(hip.eb_symbols(latent, m.entropy_bottleneck._median_vector()),
                                          m.entropy_bottleneck._median_vector(), want_f32=False, want_nhwc=True)
            with hip.KernelTimer() as kt:
                for _ in range(5):
                    m.analysis(x)
                    m.synthesis_nhwc(y_nhwc)
                torch.cuda.synchronize()
            t_enc = timeit(lambda: m.analysis(x))
            t_dec = timeit(lambda: m.synthesis_nhwc(y_nhwc))
        g = gflop_per_image(H, W)
        print(json.dumps({'shape': [N, 3, H, W], 'gflop_per_image': round(g, 3), 'encoder_ms': round(t_enc, 4),
                          'decoder_ms': round(t_dec, 4), 'tflops': round(g * N / (t_enc + t_dec), 1),
                          'frac_of_mfma_peak': round(g * N / (t_enc + t_dec) / 2500.0, 4),
                          'launches_ms': {k: round(v[1], 4) for k, v in sorted(kt.summary().items())}}))


if __name__ == '__main__':
    main()
