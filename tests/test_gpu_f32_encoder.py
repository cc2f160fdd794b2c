"""-m gpu: the reference-precision encoder mode (row g3; csrc/conv_f32.hip, `set_encoder_precision('f32')`).

north_star: "bit-identical compressed bitstream/bpp ... on the same inputs".  The byte stream of an image is a function of its
symbols round(latent - median) (sc2bench/models/layer.py:506), so identity needs the f32 latent of the reference's CPU path up
to its last bits near a rounding boundary.  Tolerances, stated here:
  * one f32 conv / GDN1 launch vs torch's CPU f32 op on the same operands: |err| <= 2e-6 * max|ref| (both are f32 sums of the
    same f32 products; only the order of additions differs);
  * symbols of the f32 encoder vs the f32 oracle encoder: mismatch rate <= 1e-4 (the bf16 encoder: ~1-2 %);
  * every image whose symbols all agree has a byte stream EQUAL to the oracle's (integer work is bit-exact)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
sys.path.insert(0, os.path.dirname(HERE))

F32_TOL = 2e-6


def _close(got, ref, tol=F32_TOL):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item() + 1e-30, 'max abs err {} vs scale {}'.format(err, ref.abs().max().item())


@pytest.mark.parametrize('cin,cout,k,s,p,hw', [(3, 96, 5, 2, 2, (37, 50)), (96, 48, 5, 2, 2, (28, 31)), (48, 24, 2, 1, 0, (13, 9)),
                                               (24, 16, 5, 2, 1, (17, 17)), (8, 200, 3, 1, 1, (10, 12)), (4, 5, 1, 1, 0, (7, 5))])
def test_conv_f32_vs_torch_cpu(S, dev, cin, cout, k, s, p, hw):
    g = torch.Generator().manual_seed(cin * 131 + cout)
    x = torch.randn(3, cin, hw[0], hw[1], generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, None, s, p)
    hip = S.hip
    xh = hip.nchw_f32_to_nhwc_f32(x.to(dev))
    assert xh.shape[-1] % 4 == 0 and torch.equal(xh[..., :cin].cpu(), x.permute(0, 2, 3, 1))
    wf = hip.pack_conv_f32(w.to(dev))
    y = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p)
    _close(y.permute(0, 3, 1, 2), ref)
    y2 = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p, out_format=hip.OUT_F32_NCHW)
    assert torch.equal(y2.cpu(), y.permute(0, 3, 1, 2).cpu())
    if cin == 3:      # the padding channel's products skipped (Kpad = 3 of Cin = 4): exact zeros left out, the same sums
        y3 = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p, cin_real=3)
        assert torch.equal(y3.cpu(), y.cpu())
        y4 = hip.conv2d_f32_fwd(x.to(dev).contiguous(), wf, cout, k, k, s, p, x_is_nchw_rgb=True)   # the NCHW image read in place
        assert torch.equal(y4.cpu(), y.cpu())
    yb = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p, epilogue=hip.EPI_BIAS, ep_beta=b.to(dev), out_format=hip.OUT_F32_NCHW)
    _close(yb, F.conv2d(x, w, b, s, p))
    med = torch.linspace(-0.4, 0.4, cout)
    sym = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p, out_format=hip.OUT_I32_NCHW_SYM, ep_beta=(med * 0.1).to(dev))
    want = torch.round(y2.cpu() * 1.0 - (med * 0.1).view(1, -1, 1, 1)).int()      # from the kernel's own f32 output: exact
    assert sym.dtype == torch.int32 and torch.equal(sym.cpu(), want)


@pytest.mark.parametrize('C,inverse', [(96, False), (48, False), (512, True), (20, False)])
def test_gdn1_f32_vs_oracle(S, R, dev, C, inverse):
    torch.manual_seed(C)
    ref = R.GDN1(C, inverse=inverse)
    with torch.no_grad():
        ref.gamma.add_(0.02 * torch.rand_like(ref.gamma))
        ref.beta.add_(0.1 * torch.rand_like(ref.beta))
    x = torch.randn(2, C, 9, 11)
    with torch.no_grad():
        want = ref(x)
    m = S.FPBasedResNetBottleneck()
    g = S.GDN1(C, inverse=inverse)
    g.load_state_dict(ref.state_dict())
    g.to(dev)
    g._tag = 't'
    gamma, beta = m._f32_pack(g)
    hip = S.hip
    xh = hip.nchw_f32_to_nhwc_f32(x.to(dev))
    y = hip.conv2d_f32_fwd(xh, gamma, C, 1, 1, 1, 0, a_op=hip.AOP_ABS, epilogue=hip.EPI_IGDN if inverse else hip.EPI_GDN,
                           ep_x=xh, ep_beta=beta, out_format=hip.OUT_F32_NCHW)
    _close(y, want, tol=4e-6)


def _pair(S, R, dev):
    from recipe import build_oracle_bottleneck
    ref, x = build_oracle_bottleneck(R)
    m = S.FPBasedResNetBottleneck()
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    m.eval().to(dev)
    m.update()
    ref.update(force=True)
    return m, ref


def test_f32_encoder_symbols_and_streams_vs_oracle(S, R, dev):
    m, ref = _pair(S, R, dev)
    g = torch.load(os.path.join(HERE, 'golden', 'fp_golden.pt'), weights_only=False)
    eb, reb = m.entropy_bottleneck, ref.entropy_bottleneck
    x = torch.rand(12, 3, 224, 224, generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        ref_latent = ref.encoder(x)
        ref_sym = reb.symbols(ref_latent)
        ref_strings = reb.compress(ref_latent)
        # default mode for comparison
        bf_sym = m.analysis(x.to(dev), symbols_for=eb).cpu()
        m.set_encoder_precision('f32')
        latent = m.analysis(x.to(dev))
        _close(latent, ref_latent, tol=1e-5)       # five chained f32 launches
        sym = m.analysis(x.to(dev), symbols_for=eb).cpu()
        assert torch.equal(sym, reb.symbols(latent.cpu())), 'fused symbol output != quantize(latent)'
        enc = m.encode(x.to(dev))
        # on the committed fixture too
        fx_sym = m.analysis(g['x'].to(dev), symbols_for=eb).cpu() if 'x' in g else None
    mism_f32 = (sym != ref_sym).float().mean().item()
    mism_bf16 = (bf_sym != ref_sym).float().mean().item()
    assert mism_f32 <= 1e-4, 'f32 encoder: symbol mismatch rate {} vs the f32 oracle'.format(mism_f32)
    assert mism_bf16 > mism_f32 or mism_bf16 == 0.0
    exact = [bool((sym[i] == ref_sym[i]).all()) for i in range(x.shape[0])]
    assert sum(exact) >= x.shape[0] // 2, 'only {} of {} images without a boundary case'.format(sum(exact), x.shape[0])
    for i, same in enumerate(exact):
        if same:
            assert enc['strings'][0][i] == ref_strings[i], 'image {}: same symbols, different bytes'.format(i)
    if fx_sym is not None:
        assert (fx_sym.reshape(-1) != g['symbols'].reshape(-1)).float().mean().item() <= 1e-4
    # the mode is a property of the model: set back, get the bf16 symbols again
    m.set_encoder_precision('bf16')
    with torch.no_grad():
        assert torch.equal(m.analysis(x.to(dev), symbols_for=eb).cpu(), bf_sym)
    with pytest.raises(ValueError):
        m.set_encoder_precision('fp8')


def test_f32_encoder_on_bench_images(S, R, dev):
    """64 images of bench.py's workload (its model, operating point and synthetic batch): mismatch <= 1e-4 against the oracle's
    f32 encoder, bpp of the device streams within 1e-4 relative of the oracle's on the same images."""
    import bench
    model = bench.build_model(dev)
    ref = bench.oracle_model(model.state_dict())
    x = bench.synthetic_batch(64, torch.device('cpu'))
    eb, reb = model.bottleneck_layer.entropy_bottleneck, ref.bottleneck_layer.entropy_bottleneck
    with torch.no_grad():
        ref_sym = reb.symbols(ref.bottleneck_layer.encoder(x))
        model.set_encoder_precision('f32')
        sym, hw = model.stage_front(x.to(dev))
        buf, off, nb, st = eb.encode_symbols_device(sym, hw[0] * hw[1])
    assert int(st.max().item()) == 0
    mism = (sym.cpu().view_as(ref_sym) != ref_sym).float().mean().item()
    assert mism <= 1e-4, 'symbol mismatch rate {}'.format(mism)
    ref_bytes = sum(len(q) for q in bench.oracle_streams(ref, ref_sym.reshape(64, -1), hw[0] * hw[1]))
    dev_bytes = int(nb.sum().item())
    assert abs(dev_bytes - ref_bytes) <= 1e-4 * ref_bytes, (dev_bytes, ref_bytes)


@pytest.mark.parametrize('cin,cout,k,s,p,inverse', [(3, 96, 5, 2, 2, False), (96, 48, 5, 2, 2, False), (8, 32, 3, 1, 1, True), (4, 20, 1, 1, 0, False)])
def test_conv_f32_fused_gdn_equals_two_launches(S, R, dev, cin, cout, k, s, p, inverse):
    """SC2_EPI_FUSED_GDN: the conv and the GDN1 over its output in one launch (the accumulators are the operand fragments of the
    1x1 GEMM over the channels) == conv launch + GDN launch, bit for bit (same products, same k order), incl. a channel count
    that is not a multiple of 16."""
    hip = S.hip
    g = torch.Generator().manual_seed(cout)
    x = torch.randn(2, cin, 21, 18, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = R.GDN1(cout, inverse=inverse)
    with torch.no_grad():
        ref.gamma.add_(0.02 * torch.rand(ref.gamma.shape, generator=g))
        ref.beta.add_(0.1 * torch.rand(ref.beta.shape, generator=g))
    gdn = S.GDN1(cout, inverse=inverse)
    gdn.load_state_dict(ref.state_dict())
    gdn.to(dev)
    gdn._tag = 't'
    gamma, beta = S.FPBasedResNetBottleneck()._f32_pack(gdn)
    xh = hip.nchw_f32_to_nhwc_f32(x.to(dev))
    wf = hip.pack_conv_f32(w.to(dev))
    t = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p)
    two = hip.conv2d_f32_fwd(t, gamma, cout, 1, 1, 1, 0, a_op=hip.AOP_ABS, epilogue=hip.EPI_IGDN if inverse else hip.EPI_GDN,
                             ep_x=t, ep_beta=beta)
    one = hip.conv2d_f32_fwd(xh, wf, cout, k, k, s, p, epilogue=hip.EPI_FUSED_IGDN if inverse else hip.EPI_FUSED_GDN,
                             ep_x=gamma, ep_beta=beta)
    assert torch.equal(one, two)
    with torch.no_grad():
        want = ref(torch.nn.functional.conv2d(x, w, None, s, p))
    _close(one.permute(0, 3, 1, 2), want, tol=4e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('inverse', [False, True])
@pytest.mark.parametrize('n,hw', [(3, (37, 50)), (2, (224, 224)), (1, (5, 7))])
def test_conv_f32_persist_equals_tile_form(S, dev, n, hw, inverse):
    """The persistent first stage (RGB planes read in place, weights + gamma + beta resident in LDS, a wave walks 32-pixel tiles:
    `conv0_gdn_f32_persist_kernel`) against the tile-per-workgroup form on the NHWC copy of the same image: the same f32 products in
    the same order -- BIT-IDENTICAL -- with ragged tile counts, odd map sizes and fewer tiles than waves."""
    hip = S.hip
    g = torch.Generator().manual_seed(n * 1000 + hw[0])
    x = torch.randn(n, 3, hw[0], hw[1], generator=g).to(dev).contiguous()
    w = (torch.randn(96, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev)
    gdn = S.GDN1(96, inverse=inverse).to(dev)
    with torch.no_grad():
        gdn.gamma.add_(0.02 * torch.rand(gdn.gamma.shape, generator=g).to(dev))
    gdn._tag = 't'
    gamma, beta = S.FPBasedResNetBottleneck()._f32_pack(gdn)
    wf = hip.pack_conv_f32(w)
    epi = hip.EPI_FUSED_IGDN if inverse else hip.EPI_FUSED_GDN
    tile = hip.conv2d_f32_fwd(hip.nchw_f32_to_nhwc_f32(x), wf, 96, 5, 5, 2, 2, epilogue=epi, ep_x=gamma, ep_beta=beta, cin_real=3)
    pers = hip.conv2d_f32_fwd(x, wf, 96, 5, 5, 2, 2, epilogue=epi, ep_x=gamma, ep_beta=beta, x_is_nchw_rgb=True)
    assert pers.shape == tile.shape and torch.isfinite(pers).all()
    assert torch.equal(pers, tile)


@pytest.mark.gpu
@pytest.mark.parametrize('cout', [64, 16, 80])
def test_conv_f32_fused_gdn_refused_for_narrow_chunks(S, R, dev, cout):
    """the fused norm GEMM walks chunk / 16 gamma k-steps, gamma is packed with ceil(Cout / 16): for Cout 49..80 or
    <= 16 the kernel would read past the gamma tensor.  The C-ABI refuses the fusion there, `conv_f32_fused_gdn_supported` says
    so, and the f32 analysis transform of a bottleneck with such a width runs as two launches and equals the oracle's."""
    hip = S.hip
    assert not hip.conv_f32_fused_gdn_supported(cout)
    assert all(hip.conv_f32_fused_gdn_supported(c) for c in (96, 48, 32, 20))
    g = torch.Generator().manual_seed(cout)
    x = torch.randn(1, 4, 9, 11, generator=g)
    w = torch.randn(cout, 4, 3, 3, generator=g) / 6.0
    gdn = S.GDN1(cout).to(dev)
    gdn._tag = 't'
    gamma, beta = S.FPBasedResNetBottleneck()._f32_pack(gdn)
    xh = hip.nchw_f32_to_nhwc_f32(x.to(dev))
    with pytest.raises(hip.Sc2Error):
        hip.conv2d_f32_fwd(xh, hip.pack_conv_f32(w.to(dev)), cout, 3, 3, 1, 1, epilogue=hip.EPI_FUSED_GDN, ep_x=gamma, ep_beta=beta)
    # the module path: conv0 of a 3 -> cout -> ... encoder takes the two-launch route and stays exact
    m = S.FPBasedResNetBottleneck()
    ref = R.FPBasedResNetBottleneck()
    conv = torch.nn.Conv2d(3, cout, 5, 2, 2, bias=False)
    enc_ref = torch.nn.Sequential(conv, R.GDN1(cout), torch.nn.Conv2d(cout, 24, 2, 1, 0, bias=False))
    enc_dev = torch.nn.Sequential(S.HipConv2d(3, cout, 5, 2, 2, bias=False), S.GDN1(cout),
                                  S.HipConv2d(cout, 24, 2, 1, 0, bias=False))
    enc_dev.load_state_dict(enc_ref.state_dict())
    for i, mod in enumerate(enc_dev):
        mod._tag = 'enc.{}'.format(i)
    m.encoder = enc_dev
    m.eval().to(dev)
    m.set_encoder_precision('f32')
    xi = torch.rand(2, 3, 32, 32, generator=g)
    with torch.no_grad():
        got = m._analysis_f32(xi.to(dev))     # (analysis() itself unpacks the five modules of the stock encoder)
        want = enc_ref(xi)
    _close(got, want, tol=4e-6)
