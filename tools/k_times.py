"""Development aid: stand-alone HIP-event timings of the bottleneck's launches (and optionally the head) at bs x 224 x 224,
the few rows an A/B of two library builds needs (SC2_LIB=tools/variants/lib_<name>.so selects the build):

    python tools/k_times.py [--bs 256] [--iters 20] [--head] [--only dec]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S  # noqa: E402
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import hip  # noqa: E402


def timeit(fn, iters=20, warmup=3):
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


def head_layer_rows(bs=256):
    """-> [(name, fn, flops)]: one representative launch per layer KIND of the task head (layer2..4 at bs x 224 x 224), each on an input of
    its own shape: conv1 / conv2 / conv3 (+ residual) of a middle block, the first block's strided layers."""
    import bench as B
    dev = torch.device('cuda:0')
    full = B.build_model(dev)
    hd = full._hip_head_for_eval()
    rows = []

    def t(n, h, w, c):
        return torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)

    def add(name, fn, flops):
        rows.append((name, fn, flops))
    N = bs
    # block indices: layer2 = 0..3, layer3 = 4..9, layer4 = 10..12
    for li, bi, hw, cmid in (('2', 1, 28, 128), ('3', 5, 14, 256), ('4', 11, 7, 512)):
        c1, c2, c3, ds = hd.blocks[bi]
        x_in, x_mid = t(N, hw, hw, 4 * cmid), t(N, hw, hw, cmid)
        m = N * hw * hw
        add('head.{}.x.c1 (1x1 {}->{})'.format(li, 4 * cmid, cmid), lambda c1=c1, x=x_in: c1(x, hip.EPI_BIAS_RELU), 2.0 * m * 4 * cmid * cmid)
        add('head.{}.x.c2 (3x3 {})'.format(li, cmid), lambda c2=c2, x=x_mid: c2(x, hip.EPI_BIAS_RELU), 2.0 * m * 9 * cmid * cmid)
        add('head.{}.x.c3 (1x1 {}->{} + res)'.format(li, cmid, 4 * cmid), lambda c3=c3, x=x_mid, r=x_in: c3(x, hip.EPI_BIAS_ADD_RELU, ep_x=r), 2.0 * m * 4 * cmid * cmid)
    for li, bi, hw, cin, cmid in (('3', 4, 28, 512, 256), ('4', 10, 14, 1024, 512)):
        c1, c2, c3, ds = hd.blocks[bi]
        x_in = t(N, hw, hw, cin)
        x_mid = t(N, hw, hw, cmid)
        mo = N * (hw // 2) * (hw // 2)
        add('head.{}.0.ds (1x1 s2 {}->{})'.format(li, cin, 4 * cmid), lambda ds=ds, x=x_in: ds(x, hip.EPI_BIAS), 2.0 * mo * cin * 4 * cmid)
        add('head.{}.0.c1 (1x1 {}->{})'.format(li, cin, cmid), lambda c1=c1, x=x_in: c1(x, hip.EPI_BIAS_RELU), 2.0 * N * hw * hw * cin * cmid)
        add('head.{}.0.c2 (3x3 s2 {})'.format(li, cmid), lambda c2=c2, x=x_mid: c2(x, hip.EPI_BIAS_RELU), 2.0 * mo * 9 * cmid * cmid)
    return rows


def build_rows(bs=256, only='', head=False, f32=False):
    """-> [(name, fn, flops)]: one callable per launch of the bottleneck forward (and optionally the head / the f32 encoder)."""
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = S.FPBasedResNetBottleneck().eval().to(dev)
    N = bs
    rows = []
    keep = []      # (tensors the callables close over stay alive with the list)

    def row(name, fn, flops):
        if only and only not in name:
            return
        rows.append((name, fn, flops))

    with torch.no_grad():
        e0, g1, e2, g3, e4 = m.encoder
        d0, h1, d2, h3, d4 = m.decoder
        x = torch.rand(N, 3, 224, 224, device=dev)
        x4 = hip.nchw_f32_to_nhwc_bf16(x, 4)
        xp = x4.view(N, 224, 112, 8)
        row('enc.nchw->nhwc4', lambda: hip.nchw_f32_to_nhwc_bf16(x, 4), 0)
        beta_g1, gamma_g1 = g1.effective_fragments()
        wf0 = m._conv0_fragments()
        a1 = hip.conv0_gdn96_fwd(xp, wf0, gamma_g1, beta_g1)
        row('enc.conv0+gdn96', lambda: hip.conv0_gdn96_fwd(xp, wf0, gamma_g1, beta_g1), (180.6e6 + 231.2e6) * N)
        row('enc.conv0+gdn96 (nchw in place)', lambda: hip.conv0_gdn96_nchw_fwd(x, wf0, gamma_g1, beta_g1), (180.6e6 + 231.2e6) * N)
        beta_g3, gamma_g3 = g3.effective()
        wq48 = e2.packed_weight(hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR)
        gf48 = hip.pack_weight_fragments(gamma_g3)
        a3 = hip.conv2_gdn48_fwd(a1, wq48, gf48, beta_g3)
        row('enc.conv2+gdn48', lambda: hip.conv2_gdn48_fwd(a1, wq48, gf48, beta_g3), (722.5e6 + 14.5e6) * N)
        row('enc.conv4', lambda: e4.forward_nhwc(a3, out_format=hip.OUT_F32_NCHW), 27.9e6 * N)
        row('enc.analysis()', lambda: m.analysis(x), 1177e6 * N)
        yh = torch.randn(N, 55, 55, 24, device=dev).to(torch.bfloat16)
        beta1, gamma1 = h1.effective_fragments()
        w0t = d0.packed_weight(hip.K_TAP_MAJOR)
        b1 = hip.conv2x2_gdn512_fwd(yh, w0t, gamma1, beta1, True)
        row('dec.conv0+igdn512', lambda: hip.conv2x2_gdn512_fwd(yh, w0t, gamma1, beta1, True), (308.3e6 + 1644.2e6) * N)
        beta3w, w2f = m._win_weights(d2, h3)
        w2p = hip.pack_conv2x2_win(d2.weight)
        row('dec.conv2 (win)', lambda: hip.conv2x2_win_fwd(b1, w2p, 0), 3171.9e6 * N)
        b3 = hip.conv2x2_win_fwd(b1, w2f, 0, beta=beta3w, inverse=True)
        row('dec.conv2+igdn256 (win)', lambda: hip.conv2x2_win_fwd(b1, w2f, 0, beta=beta3w, inverse=True), (3171.9e6 + 396.5e6) * N)
        w4p = m._win_weights(d4, None)[1]
        row('dec.conv4 (win)', lambda: hip.conv2x2_win_fwd(b3, w4p, 1), 1644.2e6 * N)
        w1 = (torch.randn(128, 256, device=dev) / 16.0).to(torch.bfloat16)
        wds = (torch.randn(512, 256, device=dev) / 16.0).to(torch.bfloat16)
        bias1, biasd = torch.randn(128, device=dev), torch.randn(512, device=dev)
        tail = hip.pack_conv2x2_win_tail(d4.weight.detach(), w1, wds)
        row('dec.conv4+head.2.0 (tail)', lambda: hip.conv2x2_win_tail_fwd(b3, tail, bias1, biasd), (1644.2e6 + 205.5e6 + 205.5e6) * N)
        m.output_format = 'bf16_nhwc'
        row('dec.synthesis()', lambda: m.synthesis_nhwc(yh), 7165e6 * N)
    if f32:
        m32 = S.FPBasedResNetBottleneck().eval().to(dev)
        m32.set_encoder_precision('f32')
        with torch.no_grad():
            x32 = torch.rand(N, 3, 224, 224, device=dev)
            row('enc.analysis() f32 operands', lambda: m32.analysis(x32), 1177e6 * N)
    if head:
        import bench as B
        full = B.build_model(dev)
        with torch.no_grad():
            feat = torch.randn(N, 56, 56, 256, device=dev).to(torch.bfloat16)
            xh = feat.permute(0, 3, 1, 2)
            row('head(hip)()', lambda: full.head(xh), 6.6e9 * N)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--head', action='store_true')
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    N = args.bs
    rows = [(name, timeit(fn, args.iters), flops) for name, fn, flops in build_rows(N, args.only, args.head)]
    print('{:<28}{:>10}{:>12}{:>10}'.format('kernel', 'ms', 'TFLOP/s', 'of 2.5PF'))
    for name, ms, flops in rows:
        print('{:<28}{:>10.4f}{:>12.1f}{:>10.3f}'.format(name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0))
    enc = [r for r in rows if r[0] == 'enc.analysis()']
    dec = [r for r in rows if r[0] == 'dec.synthesis()']
    if enc and dec:
        t = enc[0][1] + dec[0][1]
        print('bottleneck forward {:.4f} ms = {:.1f} TFLOP/s = {:.3f} of the bf16 peak'.format(t, 8.3418e9 * N / t / 1e9, 8.3418e9 * N / t / 1e9 / 2500.0))


if __name__ == '__main__':
    main()
