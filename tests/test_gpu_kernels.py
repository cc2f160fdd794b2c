"""-m gpu: per-kernel parity of the HIP path (through the C-ABI) against the CPU oracle.

Tolerances: integer / byte work (symbols given y, CDF tables, rANS streams) is bit-exact.  The MFMA kernels
take bf16 operands and accumulate in f32; they are compared with the f32 CPU op evaluated on the SAME
bf16-rounded operands, so only accumulation order and the final bf16 store rounding differ:
|err| <= 2^-8 |ref| + 1e-3 * scale for bf16 outputs, 2e-3 relative for f32 outputs.
"""
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rans as oracle_rans

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def assert_close_bf16(got, ref, what, extra=0.0):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs()
    tol = (2.0 ** -8 + extra) * ref.abs() + 2e-3 * scale
    bad = err > tol
    assert not bad.any(), '{}: {} / {} elements off, max err {} (scale {})'.format(
        what, int(bad.sum()), bad.numel(), err.max().item(), scale)


def test_layout_roundtrip(S, dev):
    x = torch.randn(3, 5, 6, 10)
    y = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), 8)
    assert y.shape == (3, 6, 10, 8) and y.dtype == torch.bfloat16
    ref = torch.zeros(3, 6, 10, 8)
    ref[..., :5] = bf16_round(x).permute(0, 2, 3, 1)
    assert torch.equal(y.float().cpu(), ref)
    y4 = S.hip.nchw_f32_to_nhwc_bf16(x[:, :3].contiguous().to(dev), 4)
    assert torch.equal(y4.float().cpu()[..., :3], bf16_round(x[:, :3]).permute(0, 2, 3, 1)) and (y4[..., 3] == 0).all()
    back = S.hip.nhwc_bf16_to_nchw_f32(y)
    assert torch.equal(back.cpu()[:, :5], bf16_round(x))


CONV_CASES = [
    # (Cin, Cout, k, stride, pad, H, W, N)   static geometries of the FP bottleneck, then generic ones
    (96, 48, 5, 2, 2, 20, 24, 2),
    (48, 24, 2, 1, 0, 9, 11, 3),
    (24, 512, 2, 1, 1, 7, 9, 2),
    (512, 256, 2, 1, 0, 8, 8, 2),      # bf16: two 256-channel halves on the window-plane kernel (runtime geometry)
    (512, 256, 2, 1, 0, 56, 56, 2),    # ... the static 55 -> 56 geometry of the 224 x 224 operating point
    (256, 256, 2, 1, 1, 7, 7, 3),
    (16, 40, 3, 1, 1, 13, 9, 2),     # generic, Cout 40 -> 48-row tile
    (64, 64, 3, 2, 1, 15, 15, 2),    # generic 64
    (32, 136, 1, 1, 0, 5, 5, 7),     # generic 128-wide, Cout not a tile multiple
    (8, 96, 3, 1, 1, 6, 6, 1),       # generic 96
    (128, 8, 1, 2, 0, 9, 9, 2),      # tiny Cout
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,H,W,N', CONV_CASES)
def test_conv_igemm(S, dev, cin, cout, k, stride, pad, H, W, N):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=stride, padding=pad)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), cin)
    wp = S.hip.pack_conv_weight(w.to(dev))
    out = S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad)
    assert out.shape == (N, ref.shape[2], ref.shape[3], cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'conv bf16 nhwc')
    out32 = S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW)
    torch.testing.assert_close(out32.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    out32h = S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NHWC)
    assert torch.equal(out32h.permute(0, 3, 1, 2).cpu(), out32.cpu())
    # weights packed [k-slab][row][32] (B tile of a slab contiguous): same result bit for bit
    wpt = S.hip.pack_conv_weight(w.to(dev), S.hip.K_B_TILE_MAJOR)
    outt = S.hip.conv2d_fwd(x_nhwc, wpt, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW, k_order=S.hip.K_B_TILE_MAJOR)
    assert torch.equal(outt.cpu(), out32.cpu())
    if cin % 32 == 0:   # slab-major K order: (channel slab, tap, channel) walk with matching weight packing
        wps = S.hip.pack_conv_weight(w.to(dev), S.hip.K_SLAB_MAJOR)
        outs = S.hip.conv2d_fwd(x_nhwc, wps, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW,
                                k_order=S.hip.K_SLAB_MAJOR)
        torch.testing.assert_close(outs.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    else:
        with pytest.raises(ValueError):
            S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad, k_order=S.hip.K_SLAB_MAJOR)
    with pytest.raises(ValueError):
        S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad, k_order=4)


@pytest.mark.parametrize('cin,cout,k,stride,pad,H,W,N', [
    (512, 256, 2, 1, 0, 9, 9, 5),      # dec.conv2 geometry (static big tile), 320 rows: 2 tiles, ragged
    (256, 256, 2, 1, 1, 8, 8, 3),      # dec.conv4 geometry
    (64, 512, 1, 1, 0, 20, 20, 1),     # generic 256-wide, 2 n-tiles, K = 64 (2 slabs < ring depth)
    (32, 128, 3, 1, 1, 17, 19, 2),     # generic 128-wide 8-wave tile
    (24, 384, 3, 2, 1, 30, 30, 2),     # Cout 384: 128-wide tiles, K = 216 (tail slab)
])
@pytest.mark.parametrize('half', ['0', '1', 'r4', 'w2'])
def test_conv_big_tile(S, dev, monkeypatch, cin, cout, k, stride, pad, H, W, N, half):
    """The 8-wave staggered 256-row kernel (forced here: its dispatch thresholds need >= 49152 rows); half = '1':
    the 128-row twins of the two decoder geometries (two workgroups per CU); 'r4': the 4-wave register-tile kernel
    (128 x 128 per wave, bf16 NHWC outputs of 256-wide tiles; SC2_CONV_BIG4 A/B variant); 'w2': one staged input
    window per channel slab for the 2x2 decoder geometries (SC2_CONV_PATCH3=2)."""
    monkeypatch.setenv('SC2_CONV_FORCE_BIG', '1')
    monkeypatch.setenv('SC2_CONV_HALF', half if half in ('0', '1') else '0')
    monkeypatch.setenv('SC2_CONV_BIG4', '1' if half == 'r4' else '0')
    monkeypatch.setenv('SC2_CONV_PATCH3', '2' if half == 'w2' else '0')   # 'w2': window-staged 2x2 decoder geometries
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=stride, padding=pad)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), cin)
    wp = S.hip.pack_conv_weight(w.to(dev))
    out32 = S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW)
    torch.testing.assert_close(out32.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    out = S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'big-tile conv bf16 nhwc')
    if cin % 32 == 0:
        order = S.hip.K_SLAB_MAJOR | S.hip.K_B_TILE_MAJOR
        wps = S.hip.pack_conv_weight(w.to(dev), order)
        outs = S.hip.conv2d_fwd(x_nhwc, wps, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW, k_order=order)
        torch.testing.assert_close(outs.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    wpt = S.hip.pack_conv_weight(w.to(dev), S.hip.K_B_TILE_MAJOR)
    outt = S.hip.conv2d_fwd(x_nhwc, wpt, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW, k_order=S.hip.K_B_TILE_MAJOR)
    assert torch.equal(outt.cpu(), out32.cpu())
    monkeypatch.delenv('SC2_CONV_FORCE_BIG')
    monkeypatch.setenv('SC2_CONV_NO_BIG', '1')
    small = S.hip.conv2d_fwd(x_nhwc, wp, cout, k, k, stride, pad, out_format=S.hip.OUT_F32_NCHW)
    torch.testing.assert_close(out32.cpu(), small.cpu(), rtol=1e-5, atol=1e-5 * ref.abs().max().item())


def test_gdn_big_tile(S, R, dev, monkeypatch):
    monkeypatch.setenv('SC2_CONV_FORCE_BIG', '1')
    for C, inverse in ((256, True), (512, True)):
        torch.manual_seed(C)
        ref_m = R.GDN1(C, inverse=inverse)
        with torch.no_grad():
            ref_m.gamma.add_(0.05 * torch.rand(C, C) / C ** 0.5)
        m = S.GDN1(C, inverse=inverse)
        m.load_state_dict(ref_m.state_dict())
        m.to(dev)
        x = torch.randn(3, C, 10, 9)
        with torch.no_grad():
            beta = ref_m.beta_reparam(ref_m.beta)
            gamma = bf16_round(ref_m.gamma_reparam(ref_m.gamma)).reshape(C, C, 1, 1)
            xb = bf16_round(x)
            norm = F.conv2d(xb.abs(), gamma, beta)
            ref = xb * norm if inverse else xb / norm
            out = m(x.to(dev))
        torch.testing.assert_close(out.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())


def test_conv_identity_asymmetric(S, dev):
    """A = I style check: 1x1 conv with a permutation-like asymmetric weight must route channels exactly."""
    cin = cout = 96
    x = torch.arange(2 * cin * 4 * 5, dtype=torch.float32).reshape(2, cin, 4, 5) % 251 - 125
    w = torch.zeros(cout, cin, 1, 1)
    for o in range(cout):
        w[o, (7 * o + 3) % cin, 0, 0] = 1.0 if o % 2 == 0 else -2.0
    ref = F.conv2d(x, w)
    out = S.hip.conv2d_fwd(S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)), S.hip.pack_conv_weight(w.to(dev)), cout, 1, 1, 1, 0,
                           out_format=S.hip.OUT_F32_NCHW)
    assert torch.equal(out.cpu(), ref)


def test_conv0_pixel_pairs(S, dev):
    """First encoder conv (3->96, k5 s2 p2) on the pixel-pair view equals the plain convolution."""
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 32, 48, generator=g)
    w = torch.randn(96, 3, 5, 5, generator=g) / 75 ** 0.5
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=2)
    x4 = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), 4)
    out = S.hip.conv2d_fwd(x4.view(2, 32, 24, 8), S.hip.pack_conv0_weight_pairs(w.to(dev)), 96, 5, 3, (2, 1), (2, 1),
                           out_format=S.hip.OUT_F32_NCHW)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())


@pytest.mark.parametrize('C,inverse', [(96, False), (48, False), (512, True), (256, True), (40, False)])
def test_gdn1(S, R, dev, C, inverse):
    torch.manual_seed(C)
    ref_m = R.GDN1(C, inverse=inverse)
    with torch.no_grad():
        ref_m.gamma.add_(0.05 * torch.rand(C, C) / C ** 0.5)
        ref_m.beta.add_(0.1 * torch.rand(C))
    m = S.GDN1(C, inverse=inverse)
    m.load_state_dict(ref_m.state_dict())
    assert sorted(m.state_dict().keys()) == sorted(ref_m.state_dict().keys())
    m.to(dev)
    x = torch.randn(2, C, 9, 7)
    with torch.no_grad():
        # oracle on the bf16-rounded operands the kernel sees (x and gamma are bf16 on the device)
        beta = ref_m.beta_reparam(ref_m.beta)
        gamma = bf16_round(ref_m.gamma_reparam(ref_m.gamma)).reshape(C, C, 1, 1)
        xb = bf16_round(x)
        norm = F.conv2d(xb.abs(), gamma, beta)
        ref = xb * norm if inverse else xb * (1.0 / norm)
        out = m(x.to(dev))
        out_b = m.forward_nhwc(S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)))
        full = ref_m(x)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    assert_close_bf16(out_b.permute(0, 3, 1, 2), ref, 'gdn bf16')
    # and against the pure-f32 oracle with the bf16 operand tolerance
    assert_close_bf16(out, full, 'gdn vs f32 oracle', extra=2.0 ** -6)


@pytest.mark.parametrize('cin,cout,k,stride,pad,inverse', [(96, 48, 5, 2, 2, False), (16, 96, 3, 1, 1, False),
                                                            (24, 64, 2, 1, 1, True), (8, 32, 1, 1, 0, False)])
def test_conv_fused_gdn(S, R, dev, cin, cout, k, stride, pad, inverse):
    """conv followed by GDN1 in ONE launch vs conv (f32) -> GDN1 of the oracle on the bf16-rounded operands."""
    torch.manual_seed(cout)
    gdn = R.GDN1(cout, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(cout, cout) / cout ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(cout))
    x = torch.randn(2, cin, 13, 10)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    with torch.no_grad():
        conv = F.conv2d(bf16_round(x), bf16_round(w), stride=stride, padding=pad)
        beta = gdn.beta_reparam(gdn.beta)
        gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
        norm = F.conv2d(bf16_round(conv).abs(), gamma.reshape(cout, cout, 1, 1), beta)   # |x| enters the MFMA as bf16
        ref = conv * norm if inverse else conv / norm
    m = S.GDN1(cout, inverse=inverse)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_d = m.effective()
    out = S.hip.conv2d_fwd(S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)), S.hip.pack_conv_weight(w.to(dev)), cout, k, k, stride,
                           pad, epilogue=S.hip.EPI_FUSED_IGDN if inverse else S.hip.EPI_FUSED_GDN, ep_x=gamma_d,
                           ep_beta=beta_d, out_format=S.hip.OUT_F32_NCHW)
    torch.testing.assert_close(out.cpu(), ref, rtol=3e-3, atol=3e-3 * ref.abs().max().item())
    with pytest.raises(S.hip.Sc2Error):   # a 256-channel GDN does not fit one tile: must be refused, not mis-computed
        S.hip.conv2d_fwd(S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)), S.hip.pack_conv_weight(torch.randn(256, cin, 1, 1).to(dev)),
                         256, 1, 1, 1, 0, epilogue=S.hip.EPI_FUSED_GDN,
                         ep_x=S.hip.pack_conv_weight(torch.eye(256).reshape(256, 256, 1, 1).to(dev)),
                         ep_beta=torch.ones(256, device=dev))


@pytest.mark.parametrize('cin,cout,H,W,N', [
    (128, 128, 28, 28, 3),     # layer2 geometry: tiles cross image rows, ragged last tile
    (64, 256, 14, 14, 9),      # layer3 geometry: a 256-pixel tile spans two images
    (32, 256, 7, 7, 11),       # layer4 geometry: five images per tile, a single channel slab
    (96, 384, 5, 31, 2),       # widest supported row, Cout 384 (128-wide tiles), three channel slabs
])
def test_conv3x3_window_kernel(S, dev, monkeypatch, cin, cout, H, W, N):
    """3x3 stride-1 pad-1 convolution with ONE staged input window per 32-channel slab serving all nine taps
    (Cfg8::PATCH3) against the f32 op on the bf16-rounded operands and against the im2col-gather kernels; image
    borders, tiles spanning images, bias + ReLU epilogue."""
    g = torch.Generator().manual_seed(cin + cout + W)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(bf16_round(x), bf16_round(w), padding=1) + bias.reshape(1, -1, 1, 1))
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), cin)
    order = S.hip.preferred_k_order(cin, 3, 3)
    assert order & S.hip.K_SLAB_MAJOR
    wp = S.hip.pack_conv_weight(w.to(dev), order)
    outs = {}
    for flag in ('1', '0', '256'):     # window kernel (128-wide tiles), im2col gather, window kernel (256-wide if it divides)
        monkeypatch.setenv('SC2_CONV_PATCH3', flag)
        outs[flag] = S.hip.conv2d_fwd(x_nhwc, wp, cout, 3, 3, 1, 1, epilogue=S.hip.EPI_BIAS_RELU, ep_beta=bias.to(dev),
                                      k_order=order)
    assert_close_bf16(outs['1'].permute(0, 3, 1, 2), ref, '3x3 window kernel')
    assert_close_bf16(outs['1'], outs['0'], 'window vs gather kernel', extra=2.0 ** -7)
    assert_close_bf16(outs['256'], outs['0'], 'window (256-wide) vs gather kernel', extra=2.0 ** -7)


@pytest.mark.parametrize('cin,cout,H,W,N', [(128, 128, 56, 56, 3), (64, 256, 29, 27, 4), (32, 512, 14, 14, 12)])
def test_conv3x3_stride2_static_tile(S, dev, monkeypatch, cin, cout, H, W, N):
    """3x3 stride-2 pad-1 convolution on the static-geometry 8-wave tile with buffer-addressed gather (out-of-image taps
    and tail rows are sent out of range and read zeros) vs the f32 op and vs the 4-wave gather kernel; odd sizes."""
    g = torch.Generator().manual_seed(cin + cout + W)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=1) + bias.reshape(1, -1, 1, 1))
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), cin)
    order = S.hip.preferred_k_order(cin, 3, 3)
    wp = S.hip.pack_conv_weight(w.to(dev), order)
    outs = {}
    for flag in ('128', '256', '0'):
        monkeypatch.setenv('SC2_CONV_S2', flag)
        outs[flag] = S.hip.conv2d_fwd(x_nhwc, wp, cout, 3, 3, 2, 1, epilogue=S.hip.EPI_BIAS_RELU, ep_beta=bias.to(dev), k_order=order)
    assert_close_bf16(outs['128'].permute(0, 3, 1, 2), ref, '3x3 stride-2 static tile')
    assert_close_bf16(outs['128'], outs['0'], 'static tile vs 4-wave gather', extra=2.0 ** -7)
    assert_close_bf16(outs['256'], outs['0'], 'static 256-wide tile vs 4-wave gather', extra=2.0 ** -7)


@pytest.mark.parametrize('half', ['0', '1', 'r4'])
@pytest.mark.parametrize('cin,k,pad,inverse', [(512, 2, 0, True), (64, 1, 0, False)])
def test_conv_fused_gdn_big_tile(S, R, dev, monkeypatch, cin, k, pad, inverse, half):
    """conv + GDN1(256) in one launch of the 256-wide 8-wave tile (x image in LDS, gamma fragments from L2), ragged
    last tile; half = '1': the 128-row twin of the dec.conv2 geometry."""
    monkeypatch.setenv('SC2_CONV_FORCE_BIG', '1')
    monkeypatch.setenv('SC2_CONV_HALF', '0' if half == 'r4' else half)
    monkeypatch.setenv('SC2_CONV_BIG4', '1' if half == 'r4' else '0')
    cout = 256
    torch.manual_seed(cin)
    gdn = R.GDN1(cout, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(cout, cout) / cout ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(cout))
    x = torch.randn(3, cin, 11, 10)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    with torch.no_grad():
        conv = F.conv2d(bf16_round(x), bf16_round(w), padding=pad)
        beta = gdn.beta_reparam(gdn.beta)
        gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
        xb = bf16_round(conv)   # x enters the second GEMM and the final multiply as bf16 (LDS image)
        norm = F.conv2d(xb.abs(), gamma.reshape(cout, cout, 1, 1), beta)
        ref = xb * norm if inverse else xb / norm
    m = S.GDN1(cout, inverse=inverse)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_d = m.effective_fragments()      # the 256-wide tile takes gamma as MFMA-fragment blocks
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv_fused_gdn_supported(tuple(x_nhwc.shape), cout, k, k, 1, pad) == 2
    for order in ((S.hip.K_TAP_MAJOR, S.hip.K_SLAB_MAJOR) if cin % 32 == 0 and k > 1 else (S.hip.K_TAP_MAJOR,)):
        out = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev), order), cout, k, k, 1, pad,
                               epilogue=S.hip.EPI_FUSED_IGDN if inverse else S.hip.EPI_FUSED_GDN, ep_x=gamma_d,
                               ep_beta=beta_d, out_format=S.hip.OUT_F32_NHWC, k_order=order)
        torch.testing.assert_close(out.permute(0, 3, 1, 2).cpu(), ref, rtol=3e-3, atol=3e-3 * ref.abs().max().item())
    outb = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, k, k, 1, pad,
                            epilogue=S.hip.EPI_FUSED_IGDN if inverse else S.hip.EPI_FUSED_GDN, ep_x=gamma_d, ep_beta=beta_d)
    assert_close_bf16(outb.permute(0, 3, 1, 2), ref, 'big-tile fused conv+gdn bf16')
    monkeypatch.delenv('SC2_CONV_FORCE_BIG')
    assert not S.hip.conv_fused_gdn_supported(tuple(x_nhwc.shape), cout, k, k, 1, pad)   # too few rows un-forced


@pytest.mark.parametrize('cin,inverse,N,H,W', [(24, True, 3, 7, 9), (24, False, 2, 5, 5), (16, True, 1, 12, 11),
                                               (8, True, 5, 9, 9), (24, True, 80, 27, 27)])
def test_conv2x2_gdn512_fused(S, R, dev, cin, inverse, N, H, W):
    """decoder[0] + decoder[1] (Conv k2 p1 -> 512, GDN1(512)) in one persistent launch vs the oracle ops on the
    bf16-rounded operands; several tiles per workgroup, ragged last tile, image borders."""
    cout = 512
    torch.manual_seed(cin + N)
    gdn = R.GDN1(cout, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(cout, cout) / cout ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(cout))
    x = torch.randn(N, cin, H, W)
    w = torch.randn(cout, cin, 2, 2) / (cin * 4) ** 0.5
    with torch.no_grad():
        conv = F.conv2d(bf16_round(x), bf16_round(w), padding=1)
        beta = gdn.beta_reparam(gdn.beta)
        gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
        tb = bf16_round(conv)   # t enters the second GEMM and the final product as bf16 (LDS image)
        norm = F.conv2d(tb.abs(), gamma.reshape(cout, cout, 1, 1), beta)
        ref = tb * norm if inverse else tb / norm
    m = S.GDN1(cout, inverse=inverse)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_d = m.effective()
    assert S.hip.conv2x2_gdn512_supported(cin, cout, 2, 2, 1, 1)
    assert not S.hip.conv2x2_gdn512_supported(cin, 256, 2, 2, 1, 1)
    assert not S.hip.conv2x2_gdn512_supported(cin, cout, 2, 2, 1, 0)
    assert not S.hip.conv2x2_gdn512_supported(32, cout, 2, 2, 1, 1)
    assert not S.hip.conv2x2_gdn512_supported(32, cout, 2, 2, 1, 1)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    _, gamma_f = m.effective_fragments()
    assert torch.equal(gamma_f[3, 5, 2 * 16 + 7].cpu(), gamma_d[3 * 16 + 7, 5 * 32 + 2 * 8:5 * 32 + 2 * 8 + 8].cpu())
    out = S.hip.conv2x2_gdn512_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), gamma_f, beta_d, inverse)
    assert out.shape == (N, H + 1, W + 1, cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'fused conv2x2 + gdn512', extra=2.0 ** -8)
    # and the same result as the two-launch path (conv -> GDN1) up to the bf16 rounding of its intermediate
    two = m.forward_nhwc(S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 2, 2, 1, 1))
    assert_close_bf16(out, two, 'fused vs two launches', extra=2.0 ** -7)
    # round 5 (training forward): with `t_out` the launch also writes the conv output in front of the GDN -- y bit-identical, t the
    # bf16 image the normalisation was computed from (every row of the ragged last tile included, nothing past it)
    for _ in range(2):
        y2, t = S.hip.conv2x2_gdn512_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), gamma_f, beta_d, inverse, want_t=True)
        assert torch.equal(y2, out) and t.shape == out.shape and bool(torch.isfinite(t.float()).all())
        assert_close_bf16(t.permute(0, 3, 1, 2), conv, 'conv output of the fused conv2x2 + gdn512')
        tf = t.float()
        n2 = torch.nn.functional.linear(tf.abs(), gamma.to(dev), beta.to(dev))
        assert_close_bf16(y2, tf * n2 if inverse else tf / n2, 'y from the emitted t', extra=2.0 ** -8)


@pytest.mark.parametrize('N,H,W,fused', [(2, 24, 20, True), (3, 22, 36, False), (1, 112, 112, True), (2, 10, 128, True)])
def test_conv5s2_patch_kernel(S, R, dev, N, H, W, fused):
    """Second encoder conv (96 -> 48, k5 s2 p2) from the LDS-resident input patch (+ fused GDN1(48)): against the f32
    op on the bf16-rounded operands, and bit-for-bit-close to the generic gather kernel; odd output heights, image
    borders, the widest supported rows (OW = 64)."""
    cin, cout = 96, 48
    torch.manual_seed(H * W)
    x = torch.randn(N, cin, H, W)
    w = torch.randn(cout, cin, 5, 5) / (cin * 25) ** 0.5
    gdn = R.GDN1(cout)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(cout, cout) / cout ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(cout))
        conv = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=2)
        ref = conv
        if fused:
            beta = gdn.beta_reparam(gdn.beta)
            gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
            ref = conv / F.conv2d(bf16_round(conv).abs(), gamma.reshape(cout, cout, 1, 1), beta)
    m = S.GDN1(cout)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_d = m.effective()
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    epi = S.hip.EPI_FUSED_GDN if fused else S.hip.EPI_NONE
    assert S.hip.conv_patch_supported(tuple(x_nhwc.shape), cout, 5, 5, 2, 2, epilogue=epi)
    assert not S.hip.conv_patch_supported(tuple(x_nhwc.shape), cout, 5, 5, 2, 2, out_format=S.hip.OUT_F32_NCHW)
    assert not S.hip.conv_patch_supported((N, H, 2 * 70, cin), cout, 5, 5, 2, 2)          # rows wider than a tile
    order = S.hip.K_SLAB_MAJOR | S.hip.K_B_FRAG_MAJOR
    kw = dict(epilogue=epi, ep_x=gamma_d if fused else None, ep_beta=beta_d if fused else None)
    out = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev), order), cout, 5, 5, 2, 2, k_order=order, **kw)
    assert out.shape == (N, ref.shape[2], ref.shape[3], cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'patch conv', extra=2.0 ** -8 if fused else 0.0)
    gen = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 5, 5, 2, 2, **kw)
    assert_close_bf16(out, gen, 'patch vs gather kernel', extra=2.0 ** -7)
    with pytest.raises(S.hip.Sc2Error):     # fragment-major weights outside the patch geometry must be refused
        S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev), order), cout, 5, 5, 2, 2, k_order=order,
                         out_format=S.hip.OUT_F32_NCHW)


@pytest.mark.parametrize('cin,cout,stride,N,H,W,res,relu', [(128, 512, 1, 3, 9, 11, True, True), (256, 1024, 1, 2, 14, 14, True, True),
                                                           (256, 512, 2, 2, 13, 10, False, False), (128, 256, 1, 40, 20, 20, False, True),
                                                           (512, 128, 1, 5, 17, 9, False, True),      # 64 x 128 units, K = 512
                                                           (512, 256, 2, 3, 15, 15, True, True),      # two 128-channel chunks
                                                           (256, 128, 1, 7, 12, 12, False, True),     # 128 x 128 units
                                                           (128, 384, 1, 2, 10, 10, True, True),      # Cout % 256 != 0
                                                           (64, 256, 1, 5, 13, 9, True, True),        # K = 64 (layer1's conv3 / downsample)
                                                           (64, 128, 1, 3, 9, 9, False, False)])
def test_conv1x1_stream(S, dev, cin, cout, stride, N, H, W, res, relu):
    """Persistent streaming 1x1 conv (+ bias, residual, ReLU) against the f32 op on the bf16-rounded operands and
    against the generic tile kernel; several units per workgroup, ragged last pixel tile, stride-2 gather."""
    g = torch.Generator().manual_seed(cin + cout + N)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=stride) + bias.view(1, -1, 1, 1)
    r = bf16_round(torch.randn(ref.shape, generator=g)) if res else None
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    assert S.hip.conv1x1_stream_supported(cin, cout, 1, 1, stride, 0)
    assert not S.hip.conv1x1_stream_supported(32, cout, 1, 1, stride, 0)
    assert not S.hip.conv1x1_stream_supported(cin, cout, 3, 3, stride, 1)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    r_nhwc = S.hip.nchw_f32_to_nhwc_bf16(r.to(dev)) if res else None
    out = S.hip.conv1x1_stream_fwd(x_nhwc, S.hip.pack_weight_fragments(w.reshape(cout, cin).to(dev)), bias.to(dev),
                                   stride=stride, residual=r_nhwc, relu=relu)
    assert out.shape == (N, ref.shape[2], ref.shape[3], cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'streaming 1x1 conv')
    epi = S.hip.EPI_BIAS_ADD_RELU if (res and relu) else S.hip.EPI_BIAS_RELU if relu else S.hip.EPI_BIAS
    gen = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 1, 1, stride, 0, epilogue=epi, ep_x=r_nhwc,
                           ep_beta=bias.to(dev))
    assert_close_bf16(out, gen, 'streaming vs tile kernel', extra=2.0 ** -8)


@pytest.mark.parametrize('N,H,W,inverse', [(3, 224, 224, False), (2, 30, 224, False), (5, 11, 224, True),
                                           (2, 9, 514, False), (1, 7, 1216, False), (2, 5, 226, True), (2, 6, 40, False),
                                           (3, 4, 2, False)])
def test_conv0_gdn96_fused(S, R, dev, N, H, W, inverse):
    """First encoder conv on pixel pairs + GDN1(96) as one persistent launch vs the f32 ops on the bf16-rounded
    operands and vs the tile kernel's fused path; image borders, odd output height, several units per workgroup; widths
    other than 224: 112-pixel output segments (513 + 1 -> 112 + 112 + 33; 1216 -> 5 x 112 + 48; 226 -> 112 + 1;
    40 -> one partial segment of 20; a single pixel pair)."""
    torch.manual_seed(H)
    x = torch.rand(N, 3, H, W) * 2 - 1
    w = torch.randn(96, 3, 5, 5) / 75 ** 0.5
    gdn = R.GDN1(96, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(96, 96) / 96 ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(96))
        conv = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=2)
        beta = gdn.beta_reparam(gdn.beta)
        gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
        norm = F.conv2d(bf16_round(conv).abs(), gamma.reshape(96, 96, 1, 1), beta)
        ref = conv * norm if inverse else conv / norm
    m = S.GDN1(96, inverse=inverse)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_f = m.effective_fragments()
    x4 = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev), 4)
    xp = x4.view(N, H, W // 2, 8)
    assert S.hip.conv0_gdn96_supported(tuple(xp.shape), 96)
    assert not S.hip.conv0_gdn96_supported((N, H, W // 2, 8), 48)
    packed = S.hip.pack_conv0_weight_pairs(w.to(dev))
    out = S.hip.conv0_gdn96_fwd(xp, S.hip.pack_weight_fragments(packed[:96]), gamma_f, beta_d, inverse)
    assert out.shape == (N, ref.shape[2], ref.shape[3], 96)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'fused conv0 + gdn96', extra=2.0 ** -8)
    _, gamma_d = m.effective()
    tile = S.hip.conv2d_fwd(xp, packed, 96, 5, 3, (2, 1), (2, 1), epilogue=S.hip.EPI_FUSED_IGDN if inverse else
                            S.hip.EPI_FUSED_GDN, ep_x=gamma_d, ep_beta=beta_d)
    assert_close_bf16(out, tile, 'persistent vs tile kernel', extra=2.0 ** -7)
    # round 5: the same launch on the f32 NCHW batch itself (colour planes read in place, rounded to bf16 while staged): the
    # layout pass and the pair view disappear, the result does not change by a bit
    in_place = S.hip.conv0_gdn96_nchw_fwd(x.to(dev), S.hip.pack_weight_fragments(packed[:96]), gamma_f, beta_d, inverse)
    assert torch.equal(in_place, out), 'planes read in place != layout pass + pair view'
    again = S.hip.conv0_gdn96_nchw_fwd(x.to(dev), S.hip.pack_weight_fragments(packed[:96]), gamma_f, beta_d, inverse)
    assert torch.equal(again, out)     # (the unit counter re-arms itself: a second launch claims the same units)
    # round 5 (training forward): with `t_out` the launch also writes the conv output in front of the GDN -- y bit-identical, every
    # pixel of t written once (odd output heights, partial segments), nothing outside
    for _ in range(2):
        guard = torch.full((out.numel() + 4096,), 7.0, dtype=torch.bfloat16, device=dev)
        y2, t = S.hip.conv0_gdn96_fwd(xp, S.hip.pack_weight_fragments(packed[:96]), gamma_f, beta_d, inverse, want_t=True)
        assert torch.equal(y2, out) and t.shape == out.shape and bool(torch.isfinite(t.float()).all())
        assert_close_bf16(t.permute(0, 3, 1, 2), conv, 'conv output of the fused conv0 + gdn96')
        del guard


@pytest.mark.parametrize('N,H,inverse', [(3, 112, False), (2, 30, False), (300, 3, False), (5, 9, True)])
def test_conv2_gdn48_fused(S, R, dev, N, H, inverse):
    """Second encoder conv (96 -> 48, k5 s2 p2) + GDN1(48) as one persistent launch with the weights resident in
    registers vs the f32 ops on the bf16-rounded operands and vs the LDS-patch tile kernel; image borders, odd output
    height, more units than workgroups."""
    W = 112
    torch.manual_seed(H + N)
    x = torch.randn(N, 96, H, W)
    w = torch.randn(48, 96, 5, 5) / 2400 ** 0.5
    gdn = R.GDN1(48, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(48, 48) / 48 ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(48))
        conv = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=2)
        beta = gdn.beta_reparam(gdn.beta)
        gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
        norm = F.conv2d(bf16_round(conv).abs(), gamma.reshape(48, 48, 1, 1), beta)
        ref = conv * norm if inverse else conv / norm
    m = S.GDN1(48, inverse=inverse)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_d = m.effective()
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv2_gdn48_supported(tuple(x_nhwc.shape), 48, 5, 5, 2, 2)
    assert not S.hip.conv2_gdn48_supported((N, H, W, 64), 48, 5, 5, 2, 2)
    order = S.hip.K_SLAB_MAJOR | S.hip.K_B_FRAG_MAJOR
    wp = S.hip.pack_conv_weight(w.to(dev), order)
    out = S.hip.conv2_gdn48_fwd(x_nhwc, wp, S.hip.pack_weight_fragments(gamma_d), beta_d, inverse)
    assert out.shape == (N, ref.shape[2], ref.shape[3], 48)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'fused conv2 + gdn48', extra=2.0 ** -8)
    tile = S.hip.conv2d_fwd(x_nhwc, wp, 48, 5, 5, 2, 2, epilogue=S.hip.EPI_FUSED_IGDN if inverse else S.hip.EPI_FUSED_GDN,
                            ep_x=gamma_d, ep_beta=beta_d, k_order=order)
    assert_close_bf16(out, tile, 'persistent vs patch kernel', extra=2.0 ** -7)


@pytest.mark.parametrize('cin,cout,stride,N,H,W,relu', [(1024, 256, 1, 3, 14, 14, True), (1024, 512, 1, 2, 9, 7, False),
                                                        (1024, 128, 1, 70, 14, 14, True), (1024, 2048, 2, 3, 14, 14, False),
                                                        (2048, 512, 1, 5, 7, 7, True), (2048, 64, 2, 9, 5, 6, True),
                                                        (2048, 1024, 1, 40, 7, 7, False)])
def test_conv1x1_kres(S, dev, cin, cout, stride, N, H, W, relu):
    """1x1 conv with K = 1024 / 2048 and the weights resident in registers (+ bias, ReLU, stride 2) against the f32 op on
    the bf16-rounded operands and against the tile kernel; ragged last pixel tile, one to sixteen channel chunks, more
    units than workgroups (dynamic claims per XCD and chunk)."""
    g = torch.Generator().manual_seed(cout + N)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=stride) + bias.view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    assert bool(S.hip.lib().sc2_conv1x1_kres_supported(cin, cout, stride))
    assert S.hip.conv1x1_kres_supported(cin, cout, 1, 1, stride, 0) == (cin == 1024)   # K = 2048: opt-in (no gain measured)
    assert not S.hip.conv1x1_kres_supported(512, cout, 1, 1, 1, 0)
    assert not S.hip.conv1x1_kres_supported(cin, cout, 1, 1, 3, 0)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    out = S.hip.conv1x1_kres_fwd(x_nhwc, S.hip.pack_weight_fragments(w.reshape(cout, cin).to(dev)), bias.to(dev), stride=stride,
                                 relu=relu)
    assert out.shape == (N, ref.shape[2], ref.shape[3], cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'weights-in-registers 1x1 conv')
    gen = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 1, 1, stride, 0,
                           epilogue=S.hip.EPI_BIAS_RELU if relu else S.hip.EPI_BIAS, ep_beta=bias.to(dev))
    assert_close_bf16(out, gen, 'kres vs tile kernel', extra=2.0 ** -8)


@pytest.mark.parametrize('cin,cout,N,HW,relu', [
    (128, 128, 3, 28, True),      # layer2 conv2: four 7-row tiles per image
    (256, 256, 5, 14, True),      # layer3 conv2: one image per tile, two channel chunks
    (512, 512, 6, 7, True),       # layer4 conv2: four images per tile, ragged last tile (6 = 4 + 2)
    (64, 128, 2, 14, False),      # two slabs only, no ReLU
    (128, 384, 1, 7, True),       # a lone image in a four-image tile, three channel chunks
])
def test_conv3x3_win(S, dev, cin, cout, N, HW, relu):
    """3x3 stride-1 pad-1 conv + bias (+ ReLU) on the window-plane kernel (conv3x3_win.hip) against the f32 op on the
    bf16-rounded operands and against the implicit-GEMM tile kernel; the zero halo of every image border, tiles that end
    inside the batch, the permuted weight rows."""
    g = torch.Generator().manual_seed(cout + N + HW)
    x = torch.randn(N, cin, HW, HW, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(bf16_round(x), bf16_round(w), padding=1) + bias.view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    assert S.hip.conv3x3_win_supported(HW, HW, cin, cout, 3, 3, 1, 1)
    assert not S.hip.conv3x3_win_supported(7, 7, cin, cout, 3, 3, 2, 1)
    assert not S.hip.conv3x3_win_supported(16, 16, cin, cout, 3, 3, 1, 1)
    assert not S.hip.conv3x3_win_supported(HW, HW, 96, cout, 3, 3, 1, 1)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    out = S.hip.conv3x3_win_fwd(x_nhwc, S.hip.pack_conv3x3_win(w.to(dev)), bias.to(dev), relu=relu)
    assert out.shape == (N, HW, HW, cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'window-plane 3x3 conv')
    gen = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 3, 3, 1, 1,
                           epilogue=S.hip.EPI_BIAS_RELU if relu else S.hip.EPI_BIAS, ep_beta=bias.to(dev))
    assert_close_bf16(out, gen, 'window-plane vs tile kernel', extra=2.0 ** -8)


@pytest.mark.parametrize('M,K,N', [(256, 2048, 1000), (5, 512, 21), (130, 128, 16)])
def test_fc_and_avgpool(S, dev, M, K, N):
    """AdaptiveAvgPool2d((1,1)) + flatten + Linear on the pooled-feature kernels (layout.hip: avgpool_nhwc_kernel, fc_kernel)
    against the f32 ops on the bf16-rounded operands; ragged batch, padded output columns."""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, 7, 7, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    f32, b16 = S.hip.avgpool_nhwc(x_nhwc, want_f32=True, want_bf16=True)
    ref_pool = bf16_round(x).mean(dim=(2, 3))
    assert (f32.cpu() - ref_pool).abs().max().item() <= 1e-5 + 1e-5 * ref_pool.abs().max().item()
    assert torch.equal(b16.float().cpu(), bf16_round(f32.cpu()))
    n16 = (N + 15) // 16 * 16
    w16 = torch.zeros(n16, K)
    w16[:N] = w
    b16v = torch.zeros(n16)
    b16v[:N] = b
    out = S.hip.fc_fwd(b16, S.hip.pack_weight_fragments(w16.to(dev)), b16v.to(dev))
    assert out.shape == (M, n16)
    ref = b16.float().cpu() @ bf16_round(w).t() + b
    assert (out[:, :N].cpu() - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-4
    assert out[:, N:].abs().max().item() == 0.0 if n16 > N else True


@pytest.mark.parametrize('cout,N,H,W', [
    (24, 3, 56, 56),      # encoder[4] of the FP bottleneck at 224 x 224: 3 x 3025 pixels, ragged last 16-pixel tile
    (24, 2, 9, 7),        # a small non-square map
    (32, 1, 5, 5),        # every padded weight row in use
    (16, 5, 2, 2),        # one output pixel per image, one channel tile
])
def test_conv2x2_c48(S, dev, cout, N, H, W):
    """The streaming form of the last encoder conv (conv2x2_c48.hip): f32 latent and int32 symbols BIT-IDENTICAL to the
    implicit-GEMM tile kernel's two output formats (same k order, same MFMA operand roles), and within tolerance of the f32 op."""
    g = torch.Generator().manual_seed(cout + N + H)
    x = torch.randn(N, 48, H, W, generator=g)
    w = torch.randn(cout, 48, 2, 2, generator=g) / 192 ** 0.5 * 3.0
    med = torch.randn(cout, generator=g)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv2x2_c48_supported(tuple(x_nhwc.shape), cout, 2, 2, 1, 0)
    assert not S.hip.conv2x2_c48_supported(tuple(x_nhwc.shape), cout, 2, 2, 1, 1)
    assert not S.hip.conv2x2_c48_supported(tuple(x_nhwc.shape), 40, 2, 2, 1, 0)
    assert not S.hip.conv2x2_c48_supported((N, H, W, 64), cout, 2, 2, 1, 0)
    wf = S.hip.pack_conv2x2_c48(w.to(dev))
    lat = S.hip.conv2x2_c48_fwd(x_nhwc, wf, cout)
    sym = S.hip.conv2x2_c48_fwd(x_nhwc, wf, cout, medians=med.to(dev))
    assert lat.shape == (N, cout, H - 1, W - 1) and lat.dtype == torch.float32 and sym.dtype == torch.int32
    ref = F.conv2d(bf16_round(x), bf16_round(w))
    assert (lat.cpu() - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-4
    wp = S.hip.pack_conv_weight(w.to(dev))
    if cout % 8 == 0:
        gen = S.hip.conv2d_fwd(x_nhwc, wp, cout, 2, 2, 1, 0, out_format=S.hip.OUT_F32_NCHW)
        assert torch.equal(lat, gen), 'latent differs from the tile kernel'
        gsym = S.hip.conv2d_fwd(x_nhwc, wp, cout, 2, 2, 1, 0, out_format=S.hip.OUT_I32_NCHW_SYM, ep_beta=med.to(dev))
        assert torch.equal(sym, gsym), 'symbols differ from the tile kernel'
    assert torch.equal(sym, torch.round(lat - med.to(dev).view(1, -1, 1, 1)).to(torch.int32))
    assert sym.abs().max().item() >= 2   # (the comparison is not vacuous)


@pytest.mark.parametrize('cin,cout,N,HW,stride,res,relu', [
    (1024, 256, 3, 14, 1, False, True),    # layer3 conv1: 588 pixels = 2 full tiles + a ragged one, two channel chunks
    (2048, 512, 5, 7, 1, False, True),     # layer4 conv1: K = 2048, 245 pixels
    (512, 2048, 2, 7, 1, True, True),      # layer4 conv3 + identity + ReLU: 16 channel chunks, a single ragged tile
    (256, 1024, 1, 14, 1, True, True),     # layer3 conv3
    (1024, 2048, 3, 14, 2, False, False),  # layer4 downsample: stride 2, no ReLU
    (128, 128, 2, 28, 2, False, False),    # one loop trip (K = 128), stride 2 on an even map
    (256, 128, 1, 5, 2, False, True),      # odd map at stride 2 (5 -> 3)
])
def test_conv1x1_win(S, dev, cin, cout, N, HW, stride, res, relu):
    """1x1 conv + bias (+ residual) (+ ReLU) on the window-plane 1x1 kernel (conv1x1_win.hip) against the f32 op on the
    bf16-rounded operands and against the implicit-GEMM tile kernel: ragged last pixel tile, stride 2, the permuted weight
    rows, the residual read in the kernel's own output layout."""
    g = torch.Generator().manual_seed(cin + cout + N + HW)
    x = torch.randn(N, cin, HW, HW, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    OH = (HW - 1) // stride + 1
    r = bf16_round(torch.randn(N, cout, OH, OH, generator=g)) if res else None
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=stride) + bias.view(1, -1, 1, 1)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    assert S.hip.conv1x1_win_supported(cin, cout, 1, 1, stride, 0)
    assert not S.hip.conv1x1_win_supported(cin, cout, 1, 1, 3, 0)
    assert not S.hip.conv1x1_win_supported(cin + 64, cout, 1, 1, stride, 0)
    assert not S.hip.conv1x1_win_supported(cin, cout, 3, 3, stride, 1)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    r_nhwc = S.hip.nchw_f32_to_nhwc_bf16(r.to(dev)) if res else None
    out = S.hip.conv1x1_win_fwd(x_nhwc, S.hip.pack_conv_win(w.to(dev)), bias.to(dev), stride=stride, residual=r_nhwc, relu=relu)
    assert out.shape == (N, OH, OH, cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'window-plane 1x1 conv')
    epi = S.hip.EPI_BIAS_ADD_RELU if res else (S.hip.EPI_BIAS_RELU if relu else S.hip.EPI_BIAS)
    gen = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 1, 1, stride, 0, epilogue=epi, ep_x=r_nhwc,
                           ep_beta=bias.to(dev))
    assert_close_bf16(out, gen, 'window-plane 1x1 vs tile kernel', extra=2.0 ** -8)


@pytest.mark.parametrize('cin,cout,N,HW,relu', [
    (128, 128, 3, 56, True),      # layer2.0 conv2: 56 -> 28, four 7-row tiles per image
    (256, 256, 5, 28, True),      # layer3.0 conv2: 28 -> 14, one image per tile, two channel chunks
    (512, 512, 6, 14, True),      # layer4.0 conv2: 14 -> 7, four images per tile, ragged last tile (6 = 4 + 2)
    (32, 128, 2, 28, False),      # a single slab, no ReLU
    (96, 384, 1, 14, True),       # a lone image in a four-image tile, three slabs, three channel chunks
])
def test_conv3x3s2_win(S, dev, cin, cout, N, HW, relu):
    """3x3 stride-2 pad-1 conv + bias (+ ReLU) on the window-plane kernel's parity-class form (conv3x3_win.hip, GeoS2) against
    the f32 op on the bf16-rounded operands and against the implicit-GEMM tile kernel: the zero halo (top / left only at stride
    2), the four parity classes of the window, tiles that end inside the batch."""
    g = torch.Generator().manual_seed(cout + N + HW)
    x = torch.randn(N, cin, HW, HW, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=1) + bias.view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    assert S.hip.conv3x3_win_supported(HW, HW, cin, cout, 3, 3, 2, 1)
    assert not S.hip.conv3x3_win_supported(HW + 2, HW + 2, cin, cout, 3, 3, 2, 1)
    assert not S.hip.conv3x3_win_supported(HW, HW, cin, 64, 3, 3, 2, 1)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    out = S.hip.conv3x3_win_fwd(x_nhwc, S.hip.pack_conv3x3_win(w.to(dev)), bias.to(dev), relu=relu, stride=2)
    assert out.shape == (N, HW // 2, HW // 2, cout)
    assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'window-plane 3x3 stride-2 conv')
    gen = S.hip.conv2d_fwd(x_nhwc, S.hip.pack_conv_weight(w.to(dev)), cout, 3, 3, 2, 1,
                           epilogue=S.hip.EPI_BIAS_RELU if relu else S.hip.EPI_BIAS, ep_beta=bias.to(dev))
    assert_close_bf16(out, gen, 'window-plane stride 2 vs tile kernel', extra=2.0 ** -8)


@pytest.mark.parametrize('cin,pad,N,H,fused,inverse,run', [
    (512, 0, 3, 56, True, True, 0),       # dec.conv2 + inverse GDN1: one tile per workgroup, last row tile of an image 3 rows
    (512, 0, 3, 56, True, True, 5),       # runs of 5 tiles: the K loops of successive tiles (and images) joined
    (256, 1, 2, 55, False, True, 0),      # dec.conv4
    (256, 1, 5, 55, False, True, 7),
    (64, 0, 2, 9, False, True, 3),        # two slabs, a map that is not square (9 x 56 -> 8 x 55: two row tiles per image)
    (128, 1, 3, 6, True, False, 2),       # forward GDN1, 6 x 55 -> 7 x 56
])
def test_conv2x2_win(S, dev, monkeypatch, cin, pad, N, H, fused, inverse, run):
    """The window-plane decoder kernel (conv2x2_win.hip), plain and with the GDN1 behind it fused in, against the tile
    kernels' launches of the same layer: BIT-IDENTICAL (same operation order per output element), and against the f32 op."""
    if run:
        monkeypatch.setenv('SC2_W2_RUN', str(run))
    W = 56 if pad == 0 else 55
    g = torch.Generator().manual_seed(cin + N + H)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(256, cin, 2, 2, generator=g) / (4 * cin) ** 0.5
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv2x2_win_supported(tuple(x_nhwc.shape), 256, 2, 2, 1, pad)
    assert not S.hip.conv2x2_win_supported(tuple(x_nhwc.shape), 128, 2, 2, 1, pad)
    assert not S.hip.conv2x2_win_supported((N, H, W, cin + 32), 256, 2, 2, 1, pad)
    order = S.hip.preferred_k_order(cin, 2, 2)
    wp = S.hip.pack_conv_weight(w.to(dev), order)
    if fused:
        gdn = S.GDN1(256, inverse=inverse).to(dev)
        with torch.no_grad():
            gdn.gamma.add_(0.02 * torch.rand(256, 256, generator=g).to(dev))
        mode = S.hip.conv_fused_gdn_supported(tuple(x_nhwc.shape), 256, 2, 2, 1, pad)
        if mode:     # the tile kernel's fused launch (large batches only)
            beta, gamma = gdn.effective_fragments() if mode == 2 else gdn.effective()
            ref = S.hip.conv2d_fwd(x_nhwc, wp, 256, 2, 2, 1, pad, epilogue=S.hip.EPI_FUSED_IGDN if inverse else S.hip.EPI_FUSED_GDN,
                                   ep_x=gamma, ep_beta=beta, k_order=order)
        else:        # two launches: conv -> bf16 -> GDN1
            ref = gdn.forward_nhwc(S.hip.conv2d_fwd(x_nhwc, wp, 256, 2, 2, 1, pad, k_order=order))
        wf = S.hip.pack_conv2x2_win(w.to(dev), gdn.gamma_reparam(gdn.gamma).detach())
        out = S.hip.conv2x2_win_fwd(x_nhwc, wf, pad, beta=gdn.beta_reparam(gdn.beta).detach().float().contiguous(), inverse=inverse)
    else:
        ref = S.hip.conv2d_fwd(x_nhwc, wp, 256, 2, 2, 1, pad, k_order=order)
        out = S.hip.conv2x2_win_fwd(x_nhwc, S.hip.pack_conv2x2_win(w.to(dev)), pad)
        f32 = F.conv2d(bf16_round(x), bf16_round(w), padding=pad)
        assert_close_bf16(out.permute(0, 3, 1, 2), f32, 'window-plane 2x2 conv')
    assert out.shape == ref.shape
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
        'window-plane decoder kernel differs from the tile kernel in {} elements'.format(int((out != ref).sum()))

@pytest.mark.parametrize('cin,pad,N,H,W,fused', [
    (512, 0, 2, 17, 129, True),     # 513 x 513 input: dec.conv2 + IGDN256 at 129 x 129 -> 128 x 128 (segments 55 + 55 + 18)
    (256, 1, 2, 9, 128, False),     # ... dec.conv4 128 -> 129 (55 + 55 + 19)
    (512, 0, 1, 40, 304, True),     # 800 x 1216 input: 200 x 304 -> 199 x 303 (five whole segments + 28)
    (256, 1, 1, 21, 303, False),    # ... 303 -> 304
    (64, 0, 3, 5, 2, False),        # narrowest maps: one output column
    (64, 1, 3, 1, 1, True),         # a 1 x 1 map with padding: 2 x 2 output
    (128, 0, 2, 6, 56 + 55, True),  # exactly two whole segments
    (128, 1, 2, 7, 56, False),      # pad 1 at the width whose pad-0 form is the static geometry
])
def test_conv2x2_win_any_width(S, dev, cin, pad, N, H, W, fused):
    """The runtime-geometry instantiation of the window-plane decoder kernel (column segments of 55 output pixels; BASELINE
    configs 4 / 5) against the tile kernels' launches of the same layer: BIT-IDENTICAL, repeated launches."""
    g = torch.Generator().manual_seed(cin + 3 * H + W)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(256, cin, 2, 2, generator=g) / (4 * cin) ** 0.5
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv2x2_win_supported(tuple(x_nhwc.shape), 256, 2, 2, 1, pad)
    order = S.hip.preferred_k_order(cin, 2, 2)
    wp = S.hip.pack_conv_weight(w.to(dev), order)
    if fused:
        gdn = S.GDN1(256, inverse=True).to(dev)
        with torch.no_grad():
            gdn.gamma.add_(0.02 * torch.rand(256, 256, generator=g).to(dev))
        ref = gdn.forward_nhwc(S.hip.conv2d_fwd(x_nhwc, wp, 256, 2, 2, 1, pad, k_order=order))
        wf = S.hip.pack_conv2x2_win(w.to(dev), gdn.gamma_reparam(gdn.gamma).detach())
        beta = gdn.beta_reparam(gdn.beta).detach().float().contiguous()
        run = lambda: S.hip.conv2x2_win_fwd(x_nhwc, wf, pad, beta=beta, inverse=True)
    else:
        ref = S.hip.conv2d_fwd(x_nhwc, wp, 256, 2, 2, 1, pad, k_order=order)
        wf = S.hip.pack_conv2x2_win(w.to(dev))
        run = lambda: S.hip.conv2x2_win_fwd(x_nhwc, wf, pad)
        f32 = F.conv2d(bf16_round(x), bf16_round(w), padding=pad)
    for _ in range(2):
        out = run()
        assert out.shape == ref.shape == (N, H + 2 * pad - 1, W + 2 * pad - 1, 256)
        if fused:
            # (the two-launch reference rounds the conv output to bf16 before the GDN as the fused kernel does: same bits)
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
                '{} elements differ from conv -> IGDN1'.format(int((out != ref).sum()))
        else:
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), '{} elements differ'.format(int((out != ref).sum()))
            assert_close_bf16(out.permute(0, 3, 1, 2), f32, 'window-plane 2x2 conv, runtime geometry')


@pytest.mark.parametrize('N,run,want_y', [(3, 0, True), (2, 5, False), (5, 3, True)])
def test_conv2x2_win_tail(S, dev, monkeypatch, N, run, want_y):
    """The last decoder conv with layer2.0's conv1 (+ folded BN + ReLU) and downsample (+ folded BN, stride 2) fused behind it
    (sc2_conv2x2_win_tail_fwd) against the three separate launches: BIT-IDENTICAL, over run lengths and repeated launches."""
    if run:
        monkeypatch.setenv('SC2_W2_RUN', str(run))
    g = torch.Generator().manual_seed(N * 7 + run)
    x = torch.randn(N, 256, 55, 55, generator=g)
    w4 = torch.randn(256, 256, 2, 2, generator=g) / 32.0
    w1 = (torch.randn(128, 256, generator=g) / 16.0)
    wds = (torch.randn(512, 256, generator=g) / 16.0)
    b1, bds = torch.randn(128, generator=g), torch.randn(512, generator=g)
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv2x2_win_tail_supported(tuple(x_nhwc.shape))
    assert not S.hip.conv2x2_win_tail_supported((N, 56, 56, 256))
    h = S.hip.conv2x2_win_fwd(x_nhwc, S.hip.pack_conv2x2_win(w4.to(dev)), 1)
    ref1 = S.hip.conv1x1_stream_fwd(h, S.hip.pack_weight_fragments(w1.to(dev)), b1.to(dev), stride=1, relu=True)
    refd = S.hip.conv1x1_stream_fwd(h, S.hip.pack_weight_fragments(wds.to(dev)), bds.to(dev), stride=2, relu=False)
    stream = S.hip.pack_conv2x2_win_tail(w4.to(dev), w1.to(dev).to(torch.bfloat16), wds.to(dev).to(torch.bfloat16))
    for _ in range(3):
        o1, ods, y = S.hip.conv2x2_win_tail_fwd(x_nhwc, stream, b1.to(dev), bds.to(dev), want_y=want_y)
        assert o1.shape == ref1.shape and ods.shape == refd.shape
        assert torch.equal(o1.view(torch.int16), ref1.view(torch.int16)), 'conv1: {} elements differ'.format(int((o1 != ref1).sum()))
        assert torch.equal(ods.view(torch.int16), refd.view(torch.int16)), 'downsample: {} elements differ'.format(int((ods != refd).sum()))
        if want_y:
            assert torch.equal(y.view(torch.int16), h.view(torch.int16))
        else:
            assert y is None


def test_persistent_encoder_kernels_many_units(S, R, dev):
    """More units than resident workgroups can take statically: every workgroup of the two persistent encoder kernels
    goes through several dynamic claims (a stale claim register once made this an endless loop).  Device-only check
    against the tile kernels on the same operands."""
    N = 40
    torch.manual_seed(5)
    # first stage: 40 x 56 = 2240 units over <= 512 workgroups
    x = (torch.rand(N, 3, 224, 224) * 2 - 1).to(dev)
    w0 = (torch.randn(96, 3, 5, 5) / 75 ** 0.5).to(dev)
    g1 = S.GDN1(96).to(dev)
    beta1, gamma1_f = g1.effective_fragments()
    _, gamma1 = g1.effective()
    xp = S.hip.nchw_f32_to_nhwc_bf16(x, 4).view(N, 224, 112, 8)
    packed0 = S.hip.pack_conv0_weight_pairs(w0)
    a = S.hip.conv0_gdn96_fwd(xp, S.hip.pack_weight_fragments(packed0[:96]), gamma1_f, beta1)
    a_tile = S.hip.conv2d_fwd(xp, packed0, 96, 5, 3, (2, 1), (2, 1), epilogue=S.hip.EPI_FUSED_GDN, ep_x=gamma1, ep_beta=beta1)
    assert_close_bf16(a, a_tile, 'conv0+gdn96 persistent vs tile, 2240 units', extra=2.0 ** -7)
    # second stage: 40 x 28 = 1120 units over <= 256 workgroups
    w2 = (torch.randn(48, 96, 5, 5) / 2400 ** 0.5).to(dev)
    g3 = S.GDN1(48).to(dev)
    beta3, gamma3 = g3.effective()
    order = S.hip.K_SLAB_MAJOR | S.hip.K_B_FRAG_MAJOR
    wp = S.hip.pack_conv_weight(w2, order)
    b = S.hip.conv2_gdn48_fwd(a, wp, S.hip.pack_weight_fragments(gamma3), beta3)
    b_tile = S.hip.conv2d_fwd(a, wp, 48, 5, 5, 2, 2, epilogue=S.hip.EPI_FUSED_GDN, ep_x=gamma3, ep_beta=beta3, k_order=order)
    assert_close_bf16(b, b_tile, 'conv2+gdn48 persistent vs patch kernel, 1120 units', extra=2.0 ** -7)
    torch.cuda.synchronize()


def _golden():
    return torch.load(os.path.join(HERE, 'golden', 'fp_golden.pt'), weights_only=False)


def _device_bottleneck(S, R, dev):
    import sys
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    from recipe import build_oracle_bottleneck
    ref, x = build_oracle_bottleneck(R)
    m = S.FPBasedResNetBottleneck()
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    m.eval().to(dev)
    return m, ref, x


def test_entropy_bottleneck_forward(S, R, dev):
    g = _golden()
    m, ref, _ = _device_bottleneck(S, R, dev)
    eb = m.entropy_bottleneck
    latent = g['latent'].to(dev)
    with torch.no_grad():
        y_hat, lik = eb(latent)  # eval: round(y - median) + median
        assert torch.equal(y_hat.cpu(), g['y_hat_eval'])
        torch.testing.assert_close(lik.cpu(), g['lik_eval'], rtol=2e-4, atol=2e-7)
        y_n, lik_n = eb(latent, training=True, noise=g['noise'].to(dev))
        torch.testing.assert_close(y_n.cpu(), g['y_hat_noise'], rtol=0, atol=1e-6)
        torch.testing.assert_close(lik_n.cpu(), g['lik_noise'], rtol=2e-4, atol=2e-7)
        # fused rate: sum of -log2 p partials == BppLoss('sum') of the oracle
        _, nhwc, _, bits = S.hip.eb_forward(latent.contiguous(), eb._cached_params(), S.hip.EB_DEQUANTIZE,
                                            want_y_hat=False, want_nhwc=True, want_lik=False, want_bits=True)
        assert abs(bits.double().sum().item() - float(g['bits_eval'])) <= 1e-4 * float(g['bits_eval'])
        assert torch.equal(nhwc.float().permute(0, 3, 1, 2).cpu(), bf16_round(g['y_hat_eval']))
        # likelihood lower bound engages far in the tails
        far = torch.full((1, 24, 2, 2), 500.0, device=dev)
        _, lik_far = eb(far)
        assert torch.all(lik_far == 1e-9)
        loss = S.BppLoss('eb', 'sum')({'eb': {'output': (y_hat, lik)}})
        assert abs(loss.item() - float(g['bits_eval'])) <= 1e-3 * float(g['bits_eval'])


@pytest.mark.parametrize('N,H,W,C', [(5, 7, 7, 2048), (3, 3, 5, 24), (2, 1, 1, 4096)])
def test_avgpool_nhwc(S, dev, N, H, W, C):
    g = torch.Generator().manual_seed(C + N)
    x = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(dev)
    f32, b16 = S.hip.avgpool_nhwc(x, want_f32=True, want_bf16=True)
    ref = x.float().mean(dim=(1, 2))
    torch.testing.assert_close(f32, ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(b16, f32.to(torch.bfloat16))


@pytest.mark.parametrize('N,C,H,W', [(3, 24, 55, 55), (2, 24, 7, 9), (5, 6, 16, 16), (2, 25, 10, 10)])
def test_dequantize_layouts(S, dev, N, C, H, W):
    """symbols + medians -> f32 NCHW and bf16 NHWC in one launch: both exact (tile-transposed form for even C, ragged last
    pixel tile, a row base that is not 16-byte aligned; plane form for odd C)."""
    g = torch.Generator().manual_seed(N * C + W)
    sym = torch.randint(-40, 40, (N, C, H, W), generator=g, dtype=torch.int32).to(dev)
    med = torch.randn(C, generator=g).to(dev)
    f32, nhwc = S.hip.eb_dequantize(sym, med, want_f32=True, want_nhwc=True)
    ref = sym.float() + med.view(1, C, 1, 1)
    assert torch.equal(f32, ref)
    assert torch.equal(nhwc, ref.permute(0, 2, 3, 1).to(torch.bfloat16))
    _, only = S.hip.eb_dequantize(sym, med, want_f32=False, want_nhwc=True)
    assert torch.equal(only, nhwc)


def test_symbols_and_dequantize_bit_exact(S, R, dev):
    g = _golden()
    m, ref, _ = _device_bottleneck(S, R, dev)
    eb = m.entropy_bottleneck
    latent = g['latent'].to(dev)
    means = m._get_means(latent)
    sym = eb.quantize(latent, 'symbols', means)
    assert sym.dtype == torch.int32 and torch.equal(sym.cpu(), g['symbols'])
    deq = eb.quantize(latent, 'dequantize', means)
    assert torch.equal(deq.cpu(), g['y_hat_eval'])
    assert torch.equal(eb.dequantize(sym, means).cpu(), g['y_hat_eval'])
    # half-way cases round to even, like torch.round
    y = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 3.4999, -2.5001]).reshape(1, 1, 7, 1).repeat(1, 24, 1, 1).to(dev)
    s = S.hip.eb_symbols(y.contiguous(), torch.zeros(24, device=dev))
    assert s[0, 0, :, 0].tolist() == [0, 2, 2, 0, -2, 3, -3]
    noisy = eb.quantize(latent, 'noise')
    assert (noisy - latent).abs().max().item() <= 0.5
    with pytest.raises(ValueError):
        eb.quantize(latent, 'nope')


def _tables(dev, rows, sizes, offs):
    width = max(len(r) for r in rows)
    cdfs = torch.tensor([r + [0] * (width - len(r)) for r in rows], dtype=torch.int32, device=dev)
    return cdfs, torch.tensor(sizes, dtype=torch.int32, device=dev), torch.tensor(offs, dtype=torch.int32, device=dev)


def _streams(buf, off, nb):
    b, o, n = buf.cpu().numpy(), off.cpu().numpy(), nb.cpu().numpy()
    return [b[i, o[i]:o[i] + n[i]].tobytes() for i in range(b.shape[0])]


def test_rans_known_answers(S, dev):
    kat = json.load(open(os.path.join(HERE, 'golden', 'rans_kat.json')))
    t = kat['table']
    cdfs, sizes, offs = _tables(dev, t['cdfs'], t['cdf_sizes'], t['offsets'])
    for case in kat['cases']:
        n = len(case['symbols'])
        sym = torch.tensor([case['symbols']], dtype=torch.int32, device=dev).reshape(1, n)
        buf, off, nb, st = S.hip.rans_encode_batch(sym, cdfs, sizes, offs, index_div=max(n, 1), out_stride=256)
        assert st.item() == 0
        assert _streams(buf, off, nb)[0].hex() == case['hex']
        dec, st2 = S.hip.rans_decode_batch(buf, off, nb, n, cdfs, sizes, offs, index_div=max(n, 1))
        assert dec.cpu().tolist() == [case['symbols']]


@pytest.mark.parametrize('n_streams,n_sym', [(1, 1), (3, 17), (64, 300), (70, 1000), (130, 257)])
def test_rans_batch_bit_exact(S, dev, n_streams, n_sym):
    rng = np.random.RandomState(n_streams * 7 + n_sym)
    rows, sizes, offs = [], [], []
    for r in range(5):
        n = rng.randint(2, 30)
        p = rng.rand(n).astype(np.float32) ** 3 + 1e-5
        p /= p.sum()
        cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-int(rng.randint(0, n)))
    cdfs, d_sizes, d_offs = _tables(dev, rows, sizes, offs)
    sym = rng.randint(-12, 13, size=(n_streams, n_sym)).astype(np.int32)
    sym[rng.rand(n_streams, n_sym) < 0.01] = 5000          # multi-nibble escapes
    sym[rng.rand(n_streams, n_sym) < 0.01] = -70000
    idx = rng.randint(0, 5, size=(n_streams, n_sym)).astype(np.int32)
    d_sym, d_idx = torch.from_numpy(sym).to(dev), torch.from_numpy(idx).to(dev)
    buf, off, nb, st = S.hip.rans_encode_batch(d_sym, cdfs, d_sizes, d_offs, indexes=d_idx,
                                               out_stride=S.hip.rans_max_bytes(n_sym))
    assert int(st.max()) == 0
    got = _streams(buf, off, nb)
    h_cdfs = cdfs.cpu().numpy()
    for i in range(n_streams):
        want = oracle_rans.encode_with_indexes(sym[i], idx[i], h_cdfs, sizes, offs)
        assert got[i] == want, 'stream {} differs'.format(i)
    dec, st2 = S.hip.rans_decode_batch(buf, off, nb, n_sym, cdfs, d_sizes, d_offs, indexes=d_idx)
    assert int(st2.max()) == 0 and np.array_equal(dec.cpu().numpy(), sym)
    # implicit indexes (entropy-bottleneck layout: row = position // index_div)
    div = max(1, (n_sym + 4) // 5)
    buf, off, nb, st = S.hip.rans_encode_batch(d_sym, cdfs, d_sizes, d_offs, index_div=div,
                                               out_stride=S.hip.rans_max_bytes(n_sym))
    imp = (np.arange(n_sym) // div).astype(np.int32)
    got = _streams(buf, off, nb)
    for i in range(0, n_streams, max(1, n_streams // 8)):
        assert got[i] == oracle_rans.encode_with_indexes(sym[i], imp, h_cdfs, sizes, offs)
    dec, _ = S.hip.rans_decode_batch(buf, off, nb, n_sym, cdfs, d_sizes, d_offs, index_div=div)
    assert np.array_equal(dec.cpu().numpy(), sym)


def test_rans_overflow_flag_and_large_table(S, dev):
    rng = np.random.RandomState(3)
    # a 400-row table does not fit the LDS path -> global-memory variant must give the same bytes
    rows, sizes, offs = [], [], []
    for r in range(400):
        n = rng.randint(2, 24)
        p = rng.rand(n).astype(np.float32) + 1e-3
        p /= p.sum()
        cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-int(rng.randint(0, n)))
    cdfs, d_sizes, d_offs = _tables(dev, rows, sizes, offs)
    assert cdfs.numel() > 6144
    sym = rng.randint(-5, 30, size=(9, 500)).astype(np.int32)
    idx = rng.randint(0, 400, size=(9, 500)).astype(np.int32)
    buf, off, nb, st = S.hip.rans_encode_batch(torch.from_numpy(sym).to(dev), cdfs, d_sizes, d_offs,
                                               indexes=torch.from_numpy(idx).to(dev),
                                               out_stride=S.hip.rans_max_bytes(500))
    got = _streams(buf, off, nb)
    h = cdfs.cpu().numpy()
    for i in range(9):
        assert got[i] == oracle_rans.encode_with_indexes(sym[i], idx[i], h, sizes, offs)
    dec, _ = S.hip.rans_decode_batch(buf, off, nb, 500, cdfs, d_sizes, d_offs, indexes=torch.from_numpy(idx).to(dev))
    assert np.array_equal(dec.cpu().numpy(), sym)
    # too small a row: status flags the overflow instead of writing out of bounds
    buf, off, nb, st = S.hip.rans_encode_batch(torch.from_numpy(sym).to(dev), cdfs, d_sizes, d_offs,
                                               indexes=torch.from_numpy(idx).to(dev), out_stride=64)
    assert int(st.min()) == 1


@pytest.mark.parametrize('n_rows,max_len,label', [(200, 90, 'ragged rows packed in LDS'), (60, 1400, 'packed rows exceed LDS: global'),
                                                   (40, 600, 'four lanes per stream, rows + buckets in LDS'),
                                                   (64, 2400, 'four lanes per stream, rows do not fit LDS: exact path on the global table'),
                                                   (1, 20000, 'one very long row')])
def test_rans_ragged_tables_explicit_indexes(S, dev, n_rows, max_len, label):
    """Per-symbol CDF rows over a wide ragged table (the shape of the Gaussian conditional model): encoder with the
    global entry table, decoder with the rows packed into LDS as u16 (or its in-kernel global fallback)."""
    rng = np.random.RandomState(n_rows)
    rows, sizes, offs = [], [], []
    for r in range(n_rows):
        n = int(rng.randint(3, max_len)) if r else max_len      # row 0 pins the table width
        p = rng.rand(n).astype(np.float32) ** 3 + 1e-4
        p /= p.sum()
        cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-int(rng.randint(0, n)))
    cdfs, d_sizes, d_offs = _tables(dev, rows, sizes, offs)
    assert cdfs.numel() > 12288
    n_streams, n_sym = 70, 333
    idx = rng.randint(0, n_rows, size=(n_streams, n_sym)).astype(np.int32)
    span = np.array(sizes)[idx]
    sym = ((rng.rand(n_streams, n_sym) * (span + 6) - 3).astype(np.int64) + np.array(offs)[idx]).astype(np.int32)   # in-table and escapes
    d_idx = torch.from_numpy(idx).to(dev)
    buf, off, nb, st = S.hip.rans_encode_batch(torch.from_numpy(sym).to(dev), cdfs, d_sizes, d_offs, indexes=d_idx,
                                               out_stride=S.hip.rans_max_bytes(n_sym))
    assert int(st.max()) == 0
    got = _streams(buf, off, nb)
    h = cdfs.cpu().numpy()
    for i in (0, 1, 33, 69):
        assert got[i] == oracle_rans.encode_with_indexes(sym[i], idx[i], h, sizes, offs), label
    dec, dst = S.hip.rans_decode_batch(buf, off, nb, n_sym, cdfs, d_sizes, d_offs, indexes=d_idx)
    assert int(dst.max()) == 0 and np.array_equal(dec.cpu().numpy(), sym), label


DGRAD_CASES = [
    # (Cin, Cout, k, stride, pad, H, W, N)
    (96, 48, 5, 2, 2, 20, 24, 2),      # enc.conv2 geometry: 4 stride-parity classes (3x3, 3x2, 2x3, 2x2 sub-filters)
    (48, 24, 2, 1, 0, 9, 11, 3),
    (24, 512, 2, 1, 1, 7, 9, 2),
    (512, 256, 2, 1, 0, 8, 8, 2),      # bf16: two 256-channel halves on the window-plane kernel (runtime geometry)
    (512, 256, 2, 1, 0, 56, 56, 2),    # ... the static 55 -> 56 geometry of the 224 x 224 operating point
    (256, 256, 2, 1, 1, 7, 7, 3),
    (64, 64, 3, 2, 1, 15, 15, 2),      # odd input size with stride 2
    (128, 32, 1, 2, 0, 9, 9, 2),       # 1x1 stride 2: three of four parity classes receive no tap
    (16, 40, 3, 1, 1, 13, 9, 2),
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,H,W,N', DGRAD_CASES)
def test_conv_dgrad(S, dev, cin, cout, k, stride, pad, H, W, N):
    """Data gradient on the forward implicit-GEMM kernel (flipped sub-filters per stride-parity class)."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = torch.randn(N, cout, OH, OW, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cout * k * k) ** 0.5
    ref = torch.nn.grad.conv2d_input((N, cin, H, W), bf16_round(w), bf16_round(gy), stride=stride, padding=pad)
    gy_nhwc = S.hip.nchw_f32_to_nhwc_bf16(gy.to(dev))
    gx = S.hip.conv2d_dgrad(gy_nhwc, w.to(dev), stride, pad, (H, W), out_dtype=torch.float32)
    assert gx.shape == (N, H, W, cin)
    torch.testing.assert_close(gx.permute(0, 3, 1, 2).cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    gxb = S.hip.conv2d_dgrad(gy_nhwc, w.to(dev), stride, pad, (H, W))
    assert_close_bf16(gxb.permute(0, 3, 1, 2), ref, 'dgrad bf16')


@pytest.mark.parametrize('cin,cout,k,stride,pad,H,W,N', DGRAD_CASES + [(8, 96, 3, 1, 1, 30, 30, 4), (256, 136, 1, 1, 0, 40, 40, 3),
                                                                        # round 5, the three tile forms (64 / 128 / 256 output channels) with
                                                                        # partial channel tiles, partial k tiles and ranges of one trip
                                                                        (16, 520, 1, 1, 0, 33, 31, 2), (40, 200, 3, 1, 1, 19, 21, 3),
                                                                        (8, 8, 3, 2, 1, 37, 41, 5), (72, 64, 1, 1, 0, 11, 13, 2),
                                                                        (136, 264, 2, 1, 1, 9, 10, 2)])
def test_conv_wgrad(S, dev, cin, cout, k, stride, pad, H, W, N):
    """Weight gradient kernel (transposing LDS fragment reads, f32 atomic combine of pixel-range partials)."""
    g = torch.Generator().manual_seed(cin * 3 + cout)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, cin, H, W, generator=g)
    gy = torch.randn(N, cout, OH, OW, generator=g)
    ref = torch.nn.grad.conv2d_weight(bf16_round(x), (cout, cin, k, k), bf16_round(gy), stride=stride, padding=pad)
    dw = S.hip.conv2d_wgrad(S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)), S.hip.nchw_f32_to_nhwc_bf16(gy.to(dev)), k, k,
                            stride, pad)
    assert dw.shape == (cout, cin, k, k)
    torch.testing.assert_close(dw.cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())


@pytest.mark.parametrize('C,inverse', [(96, False), (48, False), (256, True), (40, True), (512, True), (96, True), (256, False)])
def test_gdn1_backward(S, R, dev, C, inverse):
    """GDN1 backward on the HIP kernels vs the oracle's autograd.  C = 96 and multiples of 128: the element-wise halves in the
    epilogues of the two GEMMs (sc2_gdn1_bwd_gemm, round 5) + column sums + wgrad; other widths: element-wise pre / post kernels
    around a gamma^T GEMM.  The fused form is also compared with the five-launch form it replaces (same arithmetic, the norm in f32
    instead of rounded to bf16 in between)."""
    torch.manual_seed(C + 1)
    ref_m = R.GDN1(C, inverse=inverse)
    with torch.no_grad():
        ref_m.gamma.add_(0.05 * torch.rand(C, C) / C ** 0.5)
        ref_m.beta.add_(0.1 * torch.rand(C))
    x = bf16_round(torch.randn(2, C, 11, 9))
    gy = bf16_round(torch.randn(2, C, 11, 9))
    beta = ref_m.beta_reparam(ref_m.beta).detach().requires_grad_(True)
    gamma = bf16_round(ref_m.gamma_reparam(ref_m.gamma).detach()).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    norm = F.conv2d(xr.abs(), gamma.reshape(C, C, 1, 1), beta)
    y = xr * norm if inverse else xr / norm
    y.backward(gy)
    dx, d_beta, d_gamma = S.hip.gdn1_backward(S.hip.nchw_f32_to_nhwc_bf16(gy.to(dev)), S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)),
                                              beta.detach().to(dev), gamma.detach().to(dev), inverse)

    def rel(a, b):
        return ((a.float().cpu() - b).norm() / (b.norm() + 1e-20)).item()
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < 1.5e-2      # norm, d_norm and t pass through bf16
    assert rel(d_beta, beta.grad) < 1.5e-2
    assert rel(d_gamma, gamma.grad) < 1.5e-2
    if S.hip.weight_rows(C) % 128 == 0 or S.hip.weight_rows(C) == 96:
        # C = 96 / 256 / 512 ran on the resident-row kernels (gdn512_rows.hip, gdn96_strips.hip): its other form is the pair of GEMMs with fused epilogues; the
        # other fused widths: theirs is the five-launch form
        off = dict(gdn_rows=False) if C in (96, 256, 512) else dict(gdn_bwd_fused=False)
        S.hip.configure(**off)
        try:
            dx5, d_beta5, d_gamma5 = S.hip.gdn1_backward(S.hip.nchw_f32_to_nhwc_bf16(gy.to(dev)), S.hip.nchw_f32_to_nhwc_bf16(x.to(dev)),
                                                         beta.detach().to(dev), gamma.detach().to(dev), inverse)
        finally:
            S.hip.configure(**{k: True for k in off})
        assert rel(dx, dx5.float().cpu()) < 1e-2 and rel(d_beta, d_beta5.cpu()) < 1e-2 and rel(d_gamma, d_gamma5.cpu()) < 1e-2


@pytest.mark.parametrize('inverse', [True, False])
@pytest.mark.parametrize('shape', [(3, 16, 24), (2, 11, 9), (1, 1, 5)])
@pytest.mark.parametrize('C', [512, 256, 96])
def test_gdn512_rows(S, dev, C, shape, inverse):
    """gdn512_rows.hip: GDN1 / inverse GDN1 over 512 channels, forward and the whole backward in one launch each, against f32
    autograd on the bf16-rounded operands.  1 152 pixels = nine full 128-pixel tiles; 198 and 5 pixels: a partial last tile.  x holds
    exact zeros -- single elements, a whole pixel, a whole channel: sign(0) = 0, such an element's gradient is its direct term alone
    (the kernel parks it in dx in front of the second GEMM and fetches it back behind it).  C = 256: the second decoder GDN's
    geometry (two image rows per 1 KB direct-to-LDS / store instruction, eight k-steps, two accumulator columns per wave); C = 96:
    gdn96_strips.hip (every wave an independent worker on 32-pixel strips, the first encoder GDN)."""
    N, H, W = shape
    torch.manual_seed(N * 7 + H)
    x = bf16_round(torch.randn(N, H, W, C))
    x[0, 0, 1, :] = 0.0                      # a whole pixel
    x[:, :, :, 37] = 0.0                     # a whole channel
    x.view(-1)[torch.randperm(x.numel())[:200]] = 0.0
    x[0, 0, 2, 5] = -0.0
    gy = bf16_round(torch.randn(N, H, W, C))
    gamma = bf16_round(0.3 * torch.rand(C, C) / C ** 0.5 + 0.1 * torch.eye(C))      # (a second term as large as the direct one)
    beta = 1.0 + 0.1 * torch.rand(C)
    xr = x.clone().requires_grad_(True)
    norm = xr.abs().reshape(-1, C) @ gamma.t() + beta
    y_ref = xr.reshape(-1, C) * norm if inverse else xr.reshape(-1, C) / norm
    y_ref.backward(gy.reshape(-1, C))
    n = norm.detach()
    g2, x2 = gy.reshape(-1, C), x.reshape(-1, C)
    dn_ref = g2 * x2 if inverse else -(g2 / n) * x2 / n
    dd_ref = g2 * n if inverse else g2 / n
    xd, gd = x.to(torch.bfloat16).to(dev), gy.to(torch.bfloat16).to(dev)
    gf, gtf = S.hip.pack_weight_fragments(gamma.to(dev)), S.hip.pack_weight_fragments(gamma.t().contiguous().to(dev))
    bd = beta.to(dev)
    y = S.hip.gdn1_rows_fwd(xd, gf, bd, inverse)
    torch.cuda.synchronize()
    assert_close_bf16(y.reshape(-1, C), y_ref.detach(), 'gdn512 rows forward')
    # ... and the tile kernel's fused epilogue: the same products in the same order
    y_tile = S.hip.conv2d_fwd(xd, S.hip.pack_conv_weight(gamma.reshape(C, C, 1, 1).to(dev)), C, 1, 1, 1, 0, a_op=S.hip.AOP_ABS,
                              epilogue=S.hip.EPI_IGDN if inverse else S.hip.EPI_GDN, ep_x=xd, ep_beta=bd)
    assert torch.equal(y, y_tile)
    d_norm, dx, d_beta = S.hip.gdn1_rows_bwd(xd, gd, gf, gtf, bd, inverse, want_d_beta=True)
    torch.cuda.synchronize()
    assert_close_bf16(d_norm.reshape(-1, C), dn_ref, 'gdn512 rows d_norm')
    # d_beta = the column sums of d_norm (inside the launch for C = 96 / 256, a column-sum launch behind it for 512)
    want_db = dn_ref.sum(0)
    assert ((d_beta.cpu() - want_db).norm() / (want_db.norm() + 1e-12)).item() < 1e-2

    def rel(a, b):
        return ((a.float().cpu() - b).norm() / (b.norm() + 1e-20)).item()
    dx_ref = xr.grad.reshape(-1, C)
    assert rel(dx.reshape(-1, C), dx_ref) < 1e-2
    zero = x2 == 0
    assert int(zero.sum()) > 200
    # where x == 0 the gradient is the direct term alone, to bf16 rounding (the second term there is of the same size: adding it
    # would be a gross error)
    got0, want0 = dx.reshape(-1, C).float().cpu()[zero], dd_ref[zero]
    assert torch.allclose(dx_ref[zero], want0, rtol=1e-6, atol=1e-7)
    assert (got0 - want0).abs().max().item() <= 2.0 ** -7 * want0.abs().max().item()
    assert ((got0 - want0).abs() <= 2.0 ** -7 * want0.abs() + 1e-6).all()


@pytest.mark.parametrize('N', [3, 40, 131])
def test_persistent_decoder_kernels_equal_per_tile_kernels(S, R, dev, N, monkeypatch):
    """conv_dec_persist.hip (deferred output stores, workgroups that walk several tiles) against the one-workgroup-per-
    tile kernel: dec.conv2 + inverse GDN1(256) and dec.conv4 are the same arithmetic in the same order -> bit-identical;
    and against the f32 ops on the bf16-rounded operands.  N = 3: fewer tiles than CUs and a partial last tile (big tile
    forced); 40 / 131: 1.9 / 6.2 tiles per workgroup."""
    torch.manual_seed(N)
    if N < 17:
        monkeypatch.setenv('SC2_CONV_FORCE_BIG', '1')
    h = (torch.randn(N, 56, 56, 512) * 0.5).to(torch.bfloat16).to(dev)
    w2 = torch.randn(256, 512, 2, 2) / 2048 ** 0.5
    gdn = R.GDN1(256, inverse=True)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(256, 256) / 256 ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(256))
    g = S.GDN1(256, inverse=True)
    g.load_state_dict(gdn.state_dict())
    g.to(dev)
    c2 = S.HipConv2d(512, 256, 2, 1, 0, bias=False)
    c4 = S.HipConv2d(256, 256, 2, 1, 1, bias=False)
    with torch.no_grad():
        c2.weight.copy_(w2)
        c4.weight.copy_(torch.randn(256, 256, 2, 2) / 1024 ** 0.5)
    c2.to(dev)
    c4.to(dev)

    def run():
        fused = S.hip.conv_fused_gdn_supported(tuple(h.shape), 256, 2, 2, 1, 0)
        assert fused == 2
        beta, gamma = g.effective_fragments()
        a = S.hip.conv2d_fwd(h, c2.packed_weight(), 256, 2, 2, 1, 0, epilogue=S.hip.EPI_FUSED_IGDN, ep_x=gamma, ep_beta=beta,
                             k_order=c2.k_order())
        b = c4.forward_nhwc(a)
        plain = c2.forward_nhwc(h)
        return a, b, plain

    monkeypatch.setenv('SC2_CONV_PERSIST', '0')
    a0, b0, p0 = run()
    for mode in ('3', '2', '1'):     # 3: one phase per slab; 2: fragment reads issued a phase early; 1: persistent, deferred stores
        monkeypatch.setenv('SC2_CONV_PERSIST', mode)
        for rep in range(3 if mode != '1' else 1):     # (a race in the read-ahead schedule would come and go)
            a1, b1, p1 = run()
            torch.cuda.synchronize()
            assert a1.shape == (N, 55, 55, 256) and b1.shape == (N, 56, 56, 256)
            assert torch.equal(a1, a0) and torch.equal(b1, b0) and torch.equal(p1, p0), 'mode {}'.format(mode)
    if N > 40:
        return          # (the f32 reference of 131 images is minutes of CPU time; equality with the per-tile kernel stands)
    with torch.no_grad():
        hf = h.float().cpu().permute(0, 3, 1, 2)
        conv = F.conv2d(hf, bf16_round(w2))
        norm = F.conv2d(bf16_round(conv).abs(), bf16_round(gdn.gamma_reparam(gdn.gamma)).reshape(256, 256, 1, 1),
                        gdn.beta_reparam(gdn.beta))
        ref = conv * norm
    assert_close_bf16(a1.permute(0, 3, 1, 2), ref, 'persistent dec.conv2 + igdn256', extra=2.0 ** -8)
    assert_close_bf16(p1.permute(0, 3, 1, 2), conv, 'persistent dec.conv2')


@pytest.mark.parametrize('n_symbols_in_row', [200, 256, 257, 700])
def test_rans_lut_decoder_row_widths(S, dev, n_symbols_in_row):
    """Implicit-index (entropy-bottleneck layout) decoder on both sides of the byte-sized lookup table: rows of up to 256
    symbols take the 64 KB uint8 table, longer rows the 128 KB uint16 one; streams equal the oracle's, decode == input."""
    rng = np.random.RandomState(n_symbols_in_row)
    rows, sizes, offs = [], [], []
    for r in range(3):
        n = n_symbols_in_row if r == 1 else int(rng.randint(5, 40))
        p = rng.rand(n).astype(np.float32) ** 2 + 1e-4
        p /= p.sum()
        cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-(n // 2))
    cdfs, d_sizes, d_offs = _tables(dev, rows, sizes, offs)
    n_streams, div = 67, 150
    n_sym = 3 * div
    sym = np.zeros((n_streams, n_sym), dtype=np.int32)
    for r in range(3):
        n = sizes[r] - 2
        sym[:, r * div:(r + 1) * div] = rng.randint(offs[r] - 2, offs[r] + n + 2, size=(n_streams, div))   # a few escapes
    d_sym = torch.from_numpy(sym).to(dev)
    buf, off, nb, st = S.hip.rans_encode_batch(d_sym, cdfs, d_sizes, d_offs, index_div=div,
                                               out_stride=S.hip.rans_max_bytes(n_sym))
    assert int(st.max()) == 0
    got = _streams(buf, off, nb)
    imp = (np.arange(n_sym) // div).astype(np.int32)
    h = cdfs.cpu().numpy()
    for i in (0, 1, 33, 66):
        assert got[i] == oracle_rans.encode_with_indexes(sym[i], imp, h, sizes, offs)
    dec, dst = S.hip.rans_decode_batch(buf, off, nb, n_sym, cdfs, d_sizes, d_offs, index_div=div)
    assert int(dst.max()) == 0 and np.array_equal(dec.cpu().numpy(), sym)


def _hostile_rows(rng, n_rows, max_len):
    rows, sizes, offs = [], [], []
    for r in range(n_rows):
        n = int(rng.randint(5, max_len))
        p = rng.rand(n).astype(np.float32) ** 2 + 1e-4
        p /= p.sum()
        cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-(n // 2))
    return rows, sizes, offs


@pytest.mark.timeout(120)
@pytest.mark.parametrize('path', ['lut (implicit indexes)', 'four lanes per stream (explicit indexes, rows in LDS)',
                                  'explicit indexes, table too large for LDS'])
def test_rans_decoders_terminate_on_hostile_streams(S, dev, path):
    """decompress() takes its bytes from the network in split computing.  A stream of 0xFF bytes makes every
    escape nibble 0xF and every renormalisation word all ones: upstream's `while (val == 15)` count loop -- and a decoder that
    restates it -- never returns.  Every device decoder must terminate on (i) an all-0xFF stream, (ii) a valid stream cut short,
    and report status bit 3 for both; valid streams in the same launch still decode exactly with status 0."""
    rng = np.random.RandomState(11)
    n_streams, n_sym = 70, 400
    if path.startswith('lut'):
        rows, sizes, offs = _hostile_rows(rng, 4, 30)
        idx, div = None, 100
    elif 'too large' in path:
        rows, sizes, offs = _hostile_rows(rng, 64, 2400)
        idx, div = rng.randint(0, 64, size=(n_streams, n_sym)).astype(np.int32), 0
    else:
        rows, sizes, offs = _hostile_rows(rng, 40, 600)
        idx, div = rng.randint(0, 40, size=(n_streams, n_sym)).astype(np.int32), 0
    cdfs, d_sizes, d_offs = _tables(dev, rows, sizes, offs)
    row_of = idx if idx is not None else np.broadcast_to((np.arange(n_sym) // div).astype(np.int32), (n_streams, n_sym))
    span = np.array(sizes)[row_of]
    sym = ((rng.rand(n_streams, n_sym) * (span + 2) - 1).astype(np.int64) + np.array(offs)[row_of]).astype(np.int32)   # a few escapes
    d_idx = None if idx is None else torch.from_numpy(idx).to(dev)
    buf, off, nb, st = S.hip.rans_encode_batch(torch.from_numpy(sym).to(dev), cdfs, d_sizes, d_offs, indexes=d_idx, index_div=div,
                                               out_stride=S.hip.rans_max_bytes(n_sym))
    assert int(st.max()) == 0
    buf, off, nb = buf.clone(), off.clone(), nb.clone()
    # stream 3: every byte 0xFF (same length); stream 5: 0xFF for 4 KB (longer than any valid stream of this size);
    # stream 9: cut to its first 16 bytes; stream 64 (second wave / block): cut to 8 bytes = the bare initial state
    o3, n3 = int(off[3]), int(nb[3])
    buf[3, o3:o3 + n3] = 255
    buf[5, :] = 255
    off[5], nb[5] = 0, buf.shape[1] // 4 * 4
    nb[9] = 16
    nb[64] = 8
    dec, dst = S.hip.rans_decode_batch(buf, off, nb, n_sym, cdfs, d_sizes, d_offs, indexes=d_idx, index_div=div)
    torch.cuda.synchronize()     # (the point of the test: this returns)
    dst = dst.cpu().numpy()
    bad = [3, 5, 9, 64]
    assert all(dst[i] & 8 for i in bad), (path, dst[bad])
    good = [i for i in range(n_streams) if i not in bad]
    assert not dst[good].any(), path
    assert np.array_equal(dec.cpu().numpy()[good], sym[good]), path


@pytest.mark.parametrize('N,H,W,N2', [(8, 28, 28, 128), (3, 28, 28, 128), (1, 5, 7, 128), (16, 28, 28, 128), (5, 28, 28, 256),
                                        (1, 5, 7, 256)])
def test_conv1x1_pair_equals_two_launches(S, dev, N, H, W, N2):
    """conv3 + bn3 + residual + ReLU of a Bottleneck block and conv1 + bn1 + ReLU of the next one in ONE launch
    (sc2_conv1x1_pair_fwd; layer2 of the ResNet-50 tail, sc2bench/models/backbone.py:235-254) against the two launches the head
    otherwise makes -- same products, same accumulation order, same epilogue order: bit-identical -- and against the f32 ops on
    the bf16-rounded operands; ragged last tile (M not a multiple of 112), several launches (persistent tile claims re-arm); N2 = 128
    (the next layer2 block) and 256 (layer3.0 behind layer2.3)."""
    hip = S.hip
    g = torch.Generator().manual_seed(N * 100 + H)
    K1, C = 128, 512
    o = torch.randn(N, H, W, K1, generator=g).to(dev).to(torch.bfloat16)
    idn = torch.randn(N, H, W, C, generator=g).to(dev).to(torch.bfloat16)
    w3 = (torch.randn(C, K1, generator=g) / K1 ** 0.5).to(dev)
    w1 = (torch.randn(N2, C, generator=g) / C ** 0.5).to(dev)
    b3 = torch.randn(C, generator=g).to(dev)
    b1 = torch.randn(N2, generator=g).to(dev)
    w3f, w1f = hip.pack_weight_fragments(w3), hip.pack_weight_fragments(w1)
    assert hip.conv1x1_pair_supported(K1, C, N2)
    h_ref = hip.conv1x1_stream_fwd(o, w3f, b3, residual=idn, relu=True)
    u_ref = hip.conv1x1_win_fwd(h_ref, hip.pack_conv_win(w1.reshape(N2, C, 1, 1)), b1, relu=True)
    for _ in range(3):
        h, u = hip.conv1x1_pair_fwd(o, w3f, b3, idn, w1f, b1)
        assert torch.equal(h, h_ref), 'h differs from the streaming kernel: max {}'.format((h.float() - h_ref.float()).abs().max().item())
        assert torch.equal(u, u_ref), 'u differs from the 1x1 kernel: max {}'.format((u.float() - u_ref.float()).abs().max().item())
    # against f32 ops on the same bf16 operands
    h32 = torch.relu(o.float().reshape(-1, K1) @ w3.to(torch.bfloat16).float().t() + b3 + idn.float().reshape(-1, C))
    assert (h.float().reshape(-1, C) - h32).abs().max().item() <= 2 ** -7 * h32.abs().max().item()
    u32 = torch.relu(h.float().reshape(-1, C) @ w1.to(torch.bfloat16).float().t() + b1)
    assert (u.float().reshape(-1, N2) - u32).abs().max().item() <= 2 ** -7 * u32.abs().max().item()


@pytest.mark.parametrize('N,H,W,inverse', [(2, 257, 257, False),    # 513 x 513 input: 257 -> 129 = 56 + 56 + 17 columns
                                           (1, 400, 608, False),    # 800 x 1216: 608 -> 304 = five segments + 24
                                           (3, 23, 167, True),      # 333 x 500 -> 167 -> 84 (odd everything), inverse form
                                           (9, 9, 7, False),        # narrower than a segment, more images than XCDs
                                           (2, 11, 113, False)])    # one column more than the static geometry
def test_conv2_gdn48_any_width(S, R, dev, N, H, W, inverse):
    """The segmented instantiation of the fused second encoder stage (conv 96 -> 48, k5 s2 p2 + GDN1(48)) at widths other than
    112, against the f32 ops on the bf16-rounded operands (as test_conv2_gdn48_fused), repeated launches."""
    torch.manual_seed(N * 1000 + H + W)
    x = torch.randn(N, 96, H, W)
    w = torch.randn(48, 96, 5, 5) / 2400 ** 0.5
    gdn = R.GDN1(48, inverse=inverse)
    with torch.no_grad():
        gdn.gamma.add_(0.05 * torch.rand(48, 48) / 48 ** 0.5)
        gdn.beta.add_(0.1 * torch.rand(48))
        conv = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=2)
        beta = gdn.beta_reparam(gdn.beta)
        gamma = bf16_round(gdn.gamma_reparam(gdn.gamma))
        norm = F.conv2d(bf16_round(conv).abs(), gamma.reshape(48, 48, 1, 1), beta)
        ref = conv * norm if inverse else conv / norm
    m = S.GDN1(48, inverse=inverse)
    m.load_state_dict(gdn.state_dict())
    m.to(dev)
    beta_d, gamma_d = m.effective()
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    assert S.hip.conv2_gdn48_supported(tuple(x_nhwc.shape), 48, 5, 5, 2, 2)
    wp = S.hip.pack_conv_weight(w.to(dev), S.hip.K_SLAB_MAJOR | S.hip.K_B_FRAG_MAJOR)
    gf = S.hip.pack_weight_fragments(gamma_d)
    for _ in range(2):
        out = S.hip.conv2_gdn48_fwd(x_nhwc, wp, gf, beta_d, inverse)
        assert out.shape == (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 48) == (N, ref.shape[2], ref.shape[3], 48)
        assert_close_bf16(out.permute(0, 3, 1, 2), ref, 'fused conv2 + gdn48, segmented', extra=2.0 ** -8)
    torch.cuda.synchronize()


@pytest.mark.parametrize('N,H,W,inverse', [(3, 112, 112, False), (2, 112, 112, True),     # the static geometry (one unit = 2 output rows)
                                           (2, 257, 257, False), (3, 23, 167, True), (9, 9, 7, False)])   # the segmented one
def test_conv2_gdn48_emits_the_conv_output(S, dev, N, H, W, inverse):
    """sc2_conv2_gdn48_fwd with `t_out` (the training forward): y is bit-identical to the launch without it, and t is the conv output
    in front of the GDN -- the f32 conv on the bf16-rounded operands, rounded to bf16 once."""
    torch.manual_seed(N * 100 + H + W)
    x = torch.randn(N, 96, H, W)
    w = torch.randn(48, 96, 5, 5) / 2400 ** 0.5
    m = S.GDN1(48, inverse=inverse)
    with torch.no_grad():
        m.gamma.add_(0.05 * torch.rand(48, 48) / 48 ** 0.5)
        m.beta.add_(0.1 * torch.rand(48))
        conv = F.conv2d(bf16_round(x), bf16_round(w), stride=2, padding=2)
    m.to(dev)
    beta_d, gamma_d = m.effective()
    x_nhwc = S.hip.nchw_f32_to_nhwc_bf16(x.to(dev))
    wp = S.hip.pack_conv_weight(w.to(dev), S.hip.K_SLAB_MAJOR | S.hip.K_B_FRAG_MAJOR)
    gf = S.hip.pack_weight_fragments(gamma_d)
    plain = S.hip.conv2_gdn48_fwd(x_nhwc, wp, gf, beta_d, inverse)
    for _ in range(2):
        y, t = S.hip.conv2_gdn48_fwd(x_nhwc, wp, gf, beta_d, inverse, want_t=True)
        assert torch.equal(y, plain)
        assert t.shape == y.shape and t.dtype == torch.bfloat16
        assert_close_bf16(t.permute(0, 3, 1, 2), conv, 'conv output of the fused conv2 + gdn48')
    torch.cuda.synchronize()


@pytest.mark.parametrize('hw,cin,cout,N', [(28, 128, 128, 3), (14, 256, 256, 5), (7, 512, 512, 9)])
def test_conv3x3_win_relu_gradient_epilogue(S, dev, hw, cin, cout, N):
    """sc2_conv3x3_win_fwd with `mask`: y = mask > 0 ? conv + bias : 0 -- bit-identical to the plain launch followed by relu_bwd."""
    torch.manual_seed(hw + N)
    x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
    w = S.hip.pack_conv3x3_win((torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5).to(dev))
    b = torch.zeros(cout, device=dev)
    mask = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16)
    mask[0, 0, :, :8] = 0.0           # zeros and negative zeros count as "not positive"
    mask[0, 1, :, :8] = -0.0
    plain = S.hip.conv3x3_win_fwd(x, w, b)
    want = S.hip.relu_bwd(plain, mask)
    got = S.hip.conv3x3_win_fwd(x, w, b, mask=mask)
    assert torch.equal(got, want)


@pytest.mark.parametrize('cin,cout,hw,N,res', [(512, 128, 28, 3, False), (2048, 512, 7, 9, False), (512, 2048, 7, 5, True)])
def test_conv1x1_win_relu_gradient_epilogue(S, dev, cin, cout, hw, N, res):
    """sc2_conv1x1_win_fwd with `mask` (and with mask + residual): bit-identical to the plain launch followed by relu_bwd."""
    torch.manual_seed(cin + N)
    x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
    w = S.hip.pack_conv_win((torch.randn(cout, cin, 1, 1) / cin ** 0.5).to(dev))
    b = torch.zeros(cout, device=dev)
    mask = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16)
    r = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16) if res else None
    plain = S.hip.conv1x1_win_fwd(x, w, b, residual=r)
    want = S.hip.relu_bwd(plain, mask)
    got = S.hip.conv1x1_win_fwd(x, w, b, residual=r, mask=mask)
    assert torch.equal(got, want)


@pytest.mark.parametrize('cin,cout,hw,N,res', [(128, 512, 28, 3, True), (256, 1024, 14, 5, True), (128, 512, 9, 2, False)])
def test_conv1x1_stream_relu_gradient_store_pass(S, dev, cin, cout, hw, N, res):
    """sc2_conv1x1_stream_fwd with `mask` (the data gradient of a block's conv1 + the skip path's gradient, masked by the previous
    block's output): bit-identical to the plain launch followed by relu_bwd; several units per workgroup, ragged last tile."""
    torch.manual_seed(cin + hw)
    assert S.hip.conv1x1_stream_mask_supported(cin, cout, 1) and not S.hip.conv1x1_stream_mask_supported(512, cout, 1)
    x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
    w = S.hip.pack_weight_fragments((torch.randn(cout, cin) / cin ** 0.5).to(dev))
    b = torch.zeros(cout, device=dev)
    mask = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16)
    mask[0, 0, :, :16] = 0.0
    r = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16) if res else None
    plain = S.hip.conv1x1_stream_fwd(x, w, b, residual=r)
    want = S.hip.relu_bwd(plain, mask)
    got = S.hip.conv1x1_stream_fwd(x, w, b, residual=r, mask=mask)
    assert torch.equal(got, want)


@pytest.mark.parametrize('N,H,W,C,k,st,pd', [(3, 112, 112, 64, 3, 2, 1), (2, 17, 23, 8, 3, 2, 1), (2, 9, 9, 72, 2, 2, 0), (1, 5, 7, 16, 3, 1, 1),
                                             (2, 8, 6, 24, (3, 2), (2, 1), (1, 0))])
def test_maxpool_nhwc(S, dev, N, H, W, C, k, st, pd):
    """sc2_maxpool_nhwc against nn.functional.max_pool2d on the same bf16 map: bit for bit, NaN and the sign of zero included (the
    update rule is torch's); odd sizes, windows hanging over every border, non-square windows."""
    torch.manual_seed(H * W + C)
    x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    x[0, 0, 0, :4] = float('nan')
    x[0, 1, :, 4:8] = -0.0
    x[0, 2, :, 4:8] = 0.0
    x[N - 1, H - 1, W - 1, :] = float('-inf')
    want = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), k, st, pd).permute(0, 2, 3, 1).contiguous()
    got = S.hip.maxpool_nhwc(x, k, st, pd)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize('N,H,C,relu,res', [(4, 14, 64, True, False), (3, 7, 2048, True, True), (5, 9, 16, False, False), (6, 14, 512, True, True),
                                            (2, 28, 128, False, True), (64, 28, 512, True, True)])
def test_bn_train_kernels(S, dev, N, H, C, relu, res):
    """sc2_bn_train_fwd / _bwd (BatchNorm2d with batch statistics + residual + ReLU on a bf16 NHWC map) against torch's f32
    batch_norm + autograd on the same bf16 operands: y and dx to bf16 rounding (one rounding each), d gamma / d beta / running
    statistics to f32 accumulation order, the residual operand's gradient exactly."""
    torch.manual_seed(C + H)
    x = (torch.randn(N, H, H, C, device=dev) * 1.3 + 0.4).to(torch.bfloat16)
    r = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16) if res else None
    g = (torch.rand(C, device=dev) + 0.5).requires_grad_(True)
    b = (torch.rand(C, device=dev) - 0.5).requires_grad_(True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rm2, rv2 = rm.clone(), rv.clone()
    y, mean, rstd = S.hip.bn_train_fwd(x, g.detach(), b.detach(), rm, rv, 0.1, 1e-5, relu, residual=r)
    xf = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    rf = r.float().permute(0, 3, 1, 2).requires_grad_(True) if res else None
    z = F.batch_norm(xf, rm2, rv2, g, b, True, 0.1, 1e-5)
    if res:
        z = z + rf
    if relu:
        z = z.relu()

    def rel(a, want):
        return ((a.float() - want.float()).norm() / (want.float().norm() + 1e-12)).item()
    assert rel(y, z.permute(0, 2, 3, 1)) < 3e-3
    if N * H * H > 1:
        assert rel(rm, rm2) < 1e-5 and rel(rv, rv2) < 1e-5
    dy = torch.randn_like(x)
    z.backward(dy.float().permute(0, 3, 1, 2))
    dx, dz, dg, db = S.hip.bn_train_bwd(dy, x, y if relu else None, g.detach(), mean, rstd, want_dz=res and relu)
    # (the reference's ReLU mask comes from ITS output; where the bf16 output rounds to zero the two masks could differ: none here)
    assert rel(dx, xf.grad.permute(0, 2, 3, 1)) < 3e-3 or N * H * H == 1
    assert rel(dg, g.grad) < 1e-4 and rel(db, b.grad) < 1e-4
    if dz is not None:
        assert torch.equal(dz.float(), rf.grad.permute(0, 2, 3, 1).to(torch.bfloat16).float())


def test_clock_probe_reads_a_plausible_shader_clock(S, dev):
    """The diagnostic probe (csrc/diag.hip, tools/clock_probe.py): every probe wave reports a monotone s_memtime / s_memrealtime
    series, an XCC id below 8, and delta s_memtime / delta s_memrealtime x 100 MHz lands between 0.5 and 3 GHz on an idle chip."""
    s = S.hip.clock_probe(16, 32, 20.0)
    torch.cuda.synchronize()
    s = s.cpu()
    assert s.shape == (16, 32, 3) and int(s[:, :, 2].max()) < 8 and int(s[:, :, 2].min()) >= 0
    for w in range(16):
        t, r = s[w, :, 0], s[w, :, 1]
        assert bool((t[1:] > t[:-1]).all()) and bool((r[1:] > r[:-1]).all())
        mhz = 100.0 * float(t[-1] - t[2]) / float(r[-1] - r[2])
        assert 500.0 < mhz < 3000.0, mhz
        assert 0.9 * 29 * 2000 <= float(r[-1] - r[2]) <= 3.0 * 29 * 2000      # 29 periods of 20 us at 100 MHz
