"""Per-kernel timing of the bottleneck path at the BASELINE shape (bs x 3 x 224 x 224), HIP events on the
current stream.  Development aid: prints ms and achieved TFLOP/s (algorithmic FLOPs) or GB/s per launch."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S  # noqa: E402
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import hip  # noqa: E402


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
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=256)
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = S.FPBasedResNetBottleneck().eval().to(dev)
    from oracle import cpu_ref as R
    R.perturb_quantiles(m.entropy_bottleneck)
    m.update()
    N = args.bs
    x = torch.rand(N, 3, 224, 224, device=dev)
    rows = []
    with torch.no_grad():
        e0, g1, e2, g3, e4 = m.encoder
        d0, h1, d2, h3, d4 = m.decoder
        x4 = hip.nchw_f32_to_nhwc_bf16(x, 4)
        rows.append(('nchw->nhwc4', timeit(lambda: hip.nchw_f32_to_nhwc_bf16(x, 4), args.iters), 0, x.numel() * 4 + x4.numel() * 2))
        xp = x4.view(N, 224, 112, 8)
        w0 = m._conv0_packed()
        a0 = hip.conv2d_fwd(xp, w0, 96, 5, 3, (2, 1), (2, 1))
        rows.append(('enc.conv0', timeit(lambda: hip.conv2d_fwd(xp, w0, 96, 5, 3, (2, 1), (2, 1)), args.iters), 180.6e6 * N, x4.numel() * 2 + a0.numel() * 2))
        if hip.conv0_gdn96_supported(tuple(xp.shape), 96):
            beta_g1, gamma_g1 = g1.effective_fragments()
            wf0 = m._conv0_fragments()
            # (the shipped path since round 5: the f32 NCHW planes read in place -- the PMC traffic passes of tools/pmc_round.sh see
            #  this launch; the pair-view form, which needs the layout pass above, is what rounds 2 - 4 ran)
            rows.append(('enc.conv0+gdn96 (stream, nchw in place)', timeit(lambda: hip.conv0_gdn96_nchw_fwd(x, wf0, gamma_g1, beta_g1), args.iters),
                         (180.6e6 + 231.2e6) * N, x.numel() * 4 + a0.numel() * 2))
        _, gamma_g1r = g1.effective()
        rows.append(('enc.conv0+gdn96 (tile)', timeit(lambda: hip.conv2d_fwd(xp, w0, 96, 5, 3, (2, 1), (2, 1), epilogue=hip.EPI_FUSED_GDN,
                                                                                ep_x=gamma_g1r, ep_beta=g1.effective()[0]), args.iters),
                     (180.6e6 + 231.2e6) * N, x4.numel() * 2 + a0.numel() * 2))
        a1 = g1.forward_nhwc(a0)
        rows.append(('enc.gdn96', timeit(lambda: g1.forward_nhwc(a0), args.iters), 231.2e6 * N, a0.numel() * 4 + a1.numel() * 2))
        a2 = e2.forward_nhwc(a1)
        rows.append(('enc.conv2', timeit(lambda: e2.forward_nhwc(a1), args.iters), 722.5e6 * N, a1.numel() * 2 + a2.numel() * 2))
        beta_g3, gamma_g3 = g3.effective()
        if hip.conv2_gdn48_supported(tuple(a1.shape), 48, 5, 5, 2, 2):
            wq48 = e2.packed_weight(hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR)
            gf48 = hip.pack_weight_fragments(gamma_g3)
            rows.append(('enc.conv2+gdn48 (resident)', timeit(lambda: hip.conv2_gdn48_fwd(a1, wq48, gf48, beta_g3), args.iters),
                         (722.5e6 + 14.5e6) * N, a1.numel() * 2 + a2.numel() * 2))
        for nm, order in (('enc.conv2+gdn48 (gather)', e2.k_order()), ('enc.conv2+gdn48 (patch)', hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR)):
            wq = e2.packed_weight(order)
            rows.append((nm, timeit(lambda: hip.conv2d_fwd(a1, wq, 48, 5, 5, 2, 2, epilogue=hip.EPI_FUSED_GDN, ep_x=gamma_g3,
                                                            ep_beta=beta_g3, k_order=order), args.iters),
                         (722.5e6 + 14.5e6) * N, a1.numel() * 2 + a2.numel() * 2))
        a3 = g3.forward_nhwc(a2)
        rows.append(('enc.gdn48', timeit(lambda: g3.forward_nhwc(a2), args.iters), 14.5e6 * N, a2.numel() * 4 + a3.numel() * 2))
        y = e4.forward_nhwc(a3, out_format=hip.OUT_F32_NCHW)
        rows.append(('enc.conv4', timeit(lambda: e4.forward_nhwc(a3, out_format=hip.OUT_F32_NCHW), args.iters), 27.9e6 * N, a3.numel() * 2 + y.numel() * 4))
        eb = m.entropy_bottleneck
        params = eb._cached_params()
        rows.append(('eb.forward(lik)', timeit(lambda: hip.eb_forward(y, params, hip.EB_DEQUANTIZE), args.iters), 0, y.numel() * 12))
        med = eb._median_vector()
        sym = hip.eb_symbols(y, med)
        rows.append(('eb.symbols', timeit(lambda: hip.eb_symbols(y, med), args.iters), 0, y.numel() * 8))
        cdf, cl, off = eb._tables()
        hw = y.shape[2] * y.shape[3]
        symv = sym.view(N, 24 * hw)
        buf, o, nb, st = hip.rans_encode_batch(symv, cdf, cl, off, index_div=hw)
        rows.append(('rans.encode', timeit(lambda: hip.rans_encode_batch(symv, cdf, cl, off, index_div=hw), max(2, args.iters // 3), 1), 0, sym.numel() * 4))
        print('bytes/img mean', nb.float().mean().item(), 'status', int(st.max()))
        rows.append(('rans.decode', timeit(lambda: hip.rans_decode_batch(buf, o, nb, 24 * hw, cdf, cl, off, index_div=hw), max(2, args.iters // 3), 1), 0, sym.numel() * 4))
        _, yh = hip.eb_dequantize(sym, med, want_f32=False, want_nhwc=True)
        rows.append(('eb.dequantize', timeit(lambda: hip.eb_dequantize(sym, med, want_f32=False, want_nhwc=True), args.iters), 0, sym.numel() * 6))
        b0 = d0.forward_nhwc(yh)
        rows.append(('dec.conv0', timeit(lambda: d0.forward_nhwc(yh), args.iters), 308.3e6 * N, yh.numel() * 2 + b0.numel() * 2))
        b1 = h1.forward_nhwc(b0)
        rows.append(('dec.igdn512', timeit(lambda: h1.forward_nhwc(b0), args.iters), 1644.2e6 * N, b0.numel() * 4 + b1.numel() * 2))
        if hip.conv2x2_gdn512_supported(24, 512, 2, 2, 1, 1):
            beta1, gamma1 = h1.effective_fragments()
            rows.append(('dec.conv0+igdn512', timeit(lambda: hip.conv2x2_gdn512_fwd(yh, d0.packed_weight(hip.K_TAP_MAJOR), gamma1, beta1, True), args.iters),
                         (308.3e6 + 1644.2e6) * N, yh.numel() * 2 + b1.numel() * 2))
        b2 = d2.forward_nhwc(b1)
        rows.append(('dec.conv2', timeit(lambda: d2.forward_nhwc(b1), args.iters), 3171.9e6 * N, b1.numel() * 2 + b2.numel() * 2))
        if hip.conv_fused_gdn_supported(tuple(b1.shape), 256, 2, 2, 1, 0):
            beta3, gamma3 = h3.effective_fragments()
            rows.append(('dec.conv2+igdn256', timeit(lambda: hip.conv2d_fwd(b1, d2.packed_weight(), 256, 2, 2, 1, 0, epilogue=hip.EPI_FUSED_IGDN,
                                                                              ep_x=gamma3, ep_beta=beta3, k_order=d2.k_order()), args.iters),
                         (3171.9e6 + 396.5e6) * N, b1.numel() * 2 + b2.numel() * 2))
        if hip.conv2x2_win_supported(tuple(b1.shape), 256, 2, 2, 1, 0):   # the window-plane kernels (conv2x2_win.hip)
            beta3w, w2f = m._win_weights(d2, h3)
            w2p = hip.pack_conv2x2_win(d2.weight)
            rows.append(('dec.conv2 (win)', timeit(lambda: hip.conv2x2_win_fwd(b1, w2p, 0), args.iters), 3171.9e6 * N, b1.numel() * 2 + b2.numel() * 2))
            rows.append(('dec.conv2+igdn256 (win)', timeit(lambda: hip.conv2x2_win_fwd(b1, w2f, 0, beta=beta3w, inverse=True), args.iters),
                         (3171.9e6 + 396.5e6) * N, b1.numel() * 2 + b2.numel() * 2))
        b3 = h3.forward_nhwc(b2)
        rows.append(('dec.igdn256', timeit(lambda: h3.forward_nhwc(b2), args.iters), 396.5e6 * N, b2.numel() * 4 + b3.numel() * 2))
        b4 = d4.forward_nhwc(b3)
        rows.append(('dec.conv4', timeit(lambda: d4.forward_nhwc(b3), args.iters), 1644.2e6 * N, b3.numel() * 2 + b4.numel() * 2))
        if hip.conv2x2_win_supported(tuple(b3.shape), 256, 2, 2, 1, 1):
            w4p = m._win_weights(d4, None)[1]
            rows.append(('dec.conv4 (win)', timeit(lambda: hip.conv2x2_win_fwd(b3, w4p, 1), args.iters), 1644.2e6 * N, b3.numel() * 2 + b4.numel() * 2))
        rows.append(('analysis()', timeit(lambda: m.analysis(x), args.iters), 1177e6 * N, 0))
        m.output_format = 'bf16_nhwc'
        rows.append(('synthesis()', timeit(lambda: m.synthesis_nhwc(yh), args.iters), 7165e6 * N, 0))
    # task head (layer2..fc): fused folded-BN HIP launches vs torch modules (MIOpen), bf16 NHWC
    import bench as B
    full = B.build_model(dev)
    with torch.no_grad():
        feat = torch.randn(N, 56, 56, 256, device=dev).to(torch.bfloat16)
        xh = feat.permute(0, 3, 1, 2)
        rows.append(('head(hip)()', timeit(lambda: full.head(xh), args.iters), 6.6e9 * N, 0))
        full.use_hip_head = False
        rows.append(('head(torch)()', timeit(lambda: full.head(xh), args.iters), 6.6e9 * N, 0))
    tot = 0.0
    print('{:<18}{:>10}{:>12}{:>12}'.format('kernel', 'ms', 'TFLOP/s', 'GB/s'))
    for name, ms, flops, byts in rows:
        print('{:<18}{:>10.3f}{:>12.1f}{:>12.1f}'.format(name, ms, flops / ms / 1e9, byts / ms / 1e6))
        if '()' not in name:
            tot += ms
    print('sum of kernels', tot, 'ms for bs', N)


if __name__ == '__main__':
    main()
