"""-m gpu: the path bench.py times -- SplittableResNet.stage_front / stage_coder / stage_back at bs 256, 224x224,
eight steps' symbols concatenated into ONE coder launch (2 048 streams x 72 600 symbols) -- against the oracle's
range coder on the device symbols, against the coder's own round trip, and against forward_device (the un-staged
eval forward).  Also the status codes the coder returns for symbols outside the codable range."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hip():
    import sc2bench_amd
    return sc2bench_amd.hip


@pytest.fixture(scope='module')
def bench_model(dev):
    import bench
    model = bench.build_model(dev)
    return bench, model


def test_bench_workload_is_not_degenerate(bench_model, dev):
    """bench.shape_workload: byte counts depend on the image and escape symbols are coded."""
    bench, model = bench_model
    x = bench.synthetic_batch(64, dev, seed=0)
    with torch.no_grad():
        sym, hw = model.stage_front(x)
        eb = model.bottleneck_layer.entropy_bottleneck
        buf, off, nb, st = eb.encode_symbols_device(sym, hw[0] * hw[1])
    assert int(st.max()) == 0
    nbh = nb.cpu().numpy()
    assert nbh.max() - nbh.min() > 1000, 'stream lengths do not vary: {}'.format(nbh[:8])
    v = sym.view(64, 24, -1) - eb._offset.view(1, 24, 1)
    esc = ((v < 0) | (v >= (eb._cdf_length - 2).view(1, 24, 1))).float().mean().item()
    assert 1e-6 < esc < 1e-2, 'escape fraction {}'.format(esc)
    assert sym.min().item() < -4 and sym.max().item() > 4


def test_fused_last_conv_symbols_equal_unfused(bench_model, dev):
    """stage_front's fused conv4 + quantisation == latent -> eb_symbols, bit for bit, at 224 and on a ragged shape."""
    bench, model = bench_model
    bl = model.bottleneck_layer
    for x in (bench.synthetic_batch(64, dev, seed=9), torch.rand(3, 3, 97, 161, device=dev) * 4 - 2):
        with torch.no_grad():
            sym, hw = model.stage_front(x)
            latent = bl.analysis(x)
            want = bl.entropy_bottleneck.symbols_device(latent)
        assert sym.dtype == torch.int32 and hw == tuple(latent.shape[-2:])
        assert torch.equal(sym, want)


def test_stage_front_writes_into_group_buffer(bench_model, dev):
    """stage_front(x, out=row block of a coder-group buffer) == stage_front(x), and touches nothing but its rows."""
    bench, model = bench_model
    xs = [bench.synthetic_batch(8, dev, seed=s) for s in (3, 4, 5)]
    with torch.no_grad():
        ref = [model.stage_front(x)[0] for x in xs]
        cols = ref[0].shape[1]
        buf = torch.full((3 * 8 + 2, cols), -12345, dtype=torch.int32, device=dev)
        for k, x in enumerate(xs):
            sym, hw = model.stage_front(x, out=buf[k * 8:(k + 1) * 8])
            assert sym.data_ptr() == buf[k * 8:(k + 1) * 8].data_ptr() and tuple(sym.shape) == (8, cols)
    assert torch.equal(buf[:24], torch.cat(ref))
    assert (buf[24:] == -12345).all()


def test_fused_decode_dequantize_equals_two_launches(bench_model, dev):
    """stage_coder(dequantized=True): the coder's last pass writes the bf16 NHWC latent == decode -> dequantize_device, bit for
    bit (ragged last pixel block: 3025 = 189 * 16 + 1; a stream count that is not a multiple of 64), and stage_back accepts it."""
    bench, model = bench_model
    eb = model.bottleneck_layer.entropy_bottleneck
    x = bench.synthetic_batch(70, dev, seed=11)
    with torch.no_grad():
        sym, hw = model.stage_front(x)
        dec, nb, st = model.stage_coder(sym, hw)
        y2, nb2, st2 = model.stage_coder(sym, hw, dequantized=True)
        assert dec.dtype == torch.int32 and y2.dtype == torch.bfloat16 and tuple(y2.shape) == (70, hw[0], hw[1], 24)
        assert torch.equal(dec, sym) and int(st.max()) == 0 and int(st2.max()) == 0 and torch.equal(nb, nb2)
        _, ref = eb.dequantize_device(dec, hw)
        assert torch.equal(y2.view(torch.int16), ref.view(torch.int16))
        assert torch.equal(model.stage_back(y2, hw), model.stage_back(dec, hw))
        # symbols alongside
        cdf, cdf_len, offset = eb._tables()
        buf, off, nbb, _ = eb.encode_symbols_device(sym, hw[0] * hw[1])
        y3, st3, sym3 = _hip().rans_decode_dequantize_batch(buf, off, nbb, sym.shape[1], cdf, cdf_len, offset, hw[0] * hw[1],
                                                               eb._median_vector(), want_symbols=True)
    if y3 is not None:
        assert torch.equal(sym3, sym) and torch.equal(y3.view(torch.int16).view(-1), ref.view(torch.int16).view(-1))


def test_bench_path_2048_streams(bench_model, dev):
    bench, model = bench_model
    bs, groups = 256, 8
    eb = model.bottleneck_layer.entropy_bottleneck
    xs, syms = [], []
    with torch.no_grad():
        for g in range(groups):
            x = bench.synthetic_batch(bs, dev, seed=g)
            sym, hw = model.stage_front(x)
            if g in (0, 5):
                xs.append((g, x))
            syms.append(sym)
        assert hw == (55, 55) and syms[0].shape == (bs, 24 * 55 * 55)
        sym_all = torch.cat(syms)                                   # [2048, 72600]
        n_hw = hw[0] * hw[1]
        # the coder exactly as bench.py launches it
        dec, nb, st = model.stage_coder(sym_all, hw)
        assert int(st.max()) == 0
        assert torch.equal(dec, sym_all), 'decode(encode(symbols)) != symbols'
        # byte identity against the oracle's coder on the same device symbols, 32 sampled streams
        buf, off, nb2, st2 = eb.encode_symbols_device(sym_all, n_hw)
        assert torch.equal(nb, nb2) and int(st2.max()) == 0
        rng = np.random.RandomState(0)
        pick = sorted(set([0, 1, 255, 256, 1023, 2047] + rng.randint(0, bs * groups, size=26).tolist()))
        ref = bench.oracle_model(model.state_dict())
        reb = ref.bottleneck_layer.entropy_bottleneck
        assert torch.equal(eb._quantized_cdf.cpu(), reb._quantized_cdf)
        assert torch.equal(eb._offset.cpu(), reb._offset) and torch.equal(eb._cdf_length.cpu(), reb._cdf_length)
        idx = torch.tensor(pick, device=dev)
        got = eb.unpack_strings(buf[idx], off[idx], nb2[idx])
        want = bench.oracle_streams(ref, sym_all[idx].cpu(), n_hw)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a == b, 'stream {} differs from the oracle coder ({} vs {} bytes)'.format(pick[i], len(a), len(b))
        assert len(set(len(s) for s in got)) > 8, 'stream lengths do not vary'
        # the staged forward == the un-staged eval forward, bit for bit (same kernels, same operands)
        for g, x in xs:
            logits = model.stage_back(dec[g * bs:(g + 1) * bs], hw)
            ref_logits, nb_ref, st_ref = model.forward_device(x)
            assert torch.equal(nb_ref, nb[g * bs:(g + 1) * bs])
            assert torch.equal(logits, ref_logits), 'stage_back(stage_coder(stage_front(x))) != forward_device(x)'
            assert torch.isfinite(logits.float()).all()


def test_bench_digest_matches_oracle(bench_model, dev):
    """The two digests bench.py prints (device streams / oracle coder on the device symbols) are equal."""
    bench, model = bench_model
    x = bench.synthetic_batch(16, dev, seed=3)
    with torch.no_grad():
        sym, hw = model.stage_front(x)
        eb = model.bottleneck_layer.entropy_bottleneck
        buf, off, nb, st = eb.encode_symbols_device(sym, hw[0] * hw[1])
    dev_streams = eb.unpack_strings(buf[:8], off[:8], nb[:8])
    ref = bench.oracle_model(model.state_dict())
    assert bench.sha256_of(dev_streams) == bench.sha256_of(bench.oracle_streams(ref, sym[:8].cpu(), hw[0] * hw[1]))


def test_rans_symbol_out_of_range_sets_status(S, dev):
    """|symbol - offset| >= 2^30 (a saturated, non-finite latent): status bit 1, the launch terminates, the other streams
    are intact; values up to 2^27 and beyond (8 raw nibbles) still round-trip exactly."""
    from oracle import rans as oracle_rans
    p = np.array([0.1, 0.2, 0.4, 0.2, 0.099, 0.001], dtype=np.float32)
    cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
    cdfs = torch.tensor([cdf], dtype=torch.int32, device=dev)
    sizes = torch.tensor([len(cdf)], dtype=torch.int32, device=dev)
    offs = torch.tensor([-2], dtype=torch.int32, device=dev)
    sym = np.zeros((4, 40), dtype=np.int32)
    sym[0, 5] = 2 ** 31 - 1           # INT_MAX as eb_symbols saturates +inf
    sym[0, 9] = -2 ** 31              # INT_MIN
    sym[1, 7] = 2 ** 26 + 12345       # 7 raw nibbles: the range upstream's nibble loop still terminates on
    sym[1, 8] = -(2 ** 26)
    sym[2, :] = np.arange(40) % 5 - 2
    sym[3, 3] = 2 ** 28 + 5           # 8 raw nibbles (upstream's `raw >> 32` is undefined there): device round trip only
    sym[3, 4] = -(2 ** 29)
    d_sym = torch.from_numpy(sym).to(dev)
    buf, off, nb, st = S.hip.rans_encode_batch(d_sym, cdfs, sizes, offs, index_div=40, out_stride=S.hip.rans_max_bytes(40))
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [2, 0, 0, 0]
    got = [b for b in S.EntropyBottleneck.unpack_strings(buf, off, nb)]
    idx = np.zeros(40, dtype=np.int32)
    for i in (1, 2):
        assert got[i] == oracle_rans.encode_with_indexes(sym[i], idx, cdfs.cpu().numpy(), [len(cdf)], [-2])
    dec, _ = S.hip.rans_decode_batch(buf, off, nb, 40, cdfs, sizes, offs, index_div=40)
    assert np.array_equal(dec.cpu().numpy()[1:], sym[1:])
    # the module API raises instead of returning a stream that does not decode to its input
    eb = S.EntropyBottleneck(1)
    eb._quantized_cdf, eb._cdf_length, eb._offset = cdfs, sizes, offs
    eb.to(dev)
    y = torch.zeros(1, 1, 4, 4, device=dev)
    y[0, 0, 1, 1] = float('inf')
    with pytest.raises(ValueError):
        eb.compress(y)


def test_bpp_estimated_against_the_oracle(bench_model, dev):
    """The line's `bpp_estimated` (eval-mode -sum log2 p / pixels from eb_forward_kernel: what BppLoss trains,
    sc2bench/loss.py:20-37) against the oracle's value on the same images: (i) the device kernel on the ORACLE'S latent equals
    the oracle's sum to 1e-5; (ii) through the reference-precision encoder the figure is the oracle's to 1e-4, through the
    bf16 encoder to 1e-2; (iii) the estimate and the rate actually coded agree within 3 % (16-bit tables, escapes)."""
    bench, model = bench_model
    x = bench.synthetic_batch(16, dev, seed=3)
    pc = bench.precision_check(model, x, dev, n=16)
    ref = pc['reference_f32_cpu']['bpp_estimated']
    assert ref > 0.5
    assert abs(pc['f32_encoder']['bpp_estimated'] - ref) <= 1e-4 * ref, (pc['f32_encoder']['bpp_estimated'], ref)
    assert abs(pc['bf16_encoder']['bpp_estimated'] - ref) <= 1e-2 * ref, (pc['bf16_encoder']['bpp_estimated'], ref)
    assert abs(pc['reference_f32_cpu']['bpp'] - ref) <= 3e-2 * ref, (pc['reference_f32_cpu']['bpp'], ref)
    oracle = bench.oracle_model(model.state_dict())
    with torch.no_grad():
        lat = oracle.bottleneck_layer.encoder(x[:4].float().cpu())
        want = float(-torch.log2(oracle.bottleneck_layer.entropy_bottleneck(lat)[1]).sum().item())
        got = float(-torch.log2(model.bottleneck_layer.entropy_bottleneck(lat.to(dev))[1].float()).sum().item())
    assert abs(got - want) <= 1e-5 * want, (got, want)


@pytest.mark.gpu
def test_workload_line_mshp224_on_the_device_coder(dev, capsys):
    """`bench.py --workload mshp224` in-process at 72 images (more streams than the host coder takes: both byte streams go
    through the batched HIP coder, the y stream through the per-symbol-index decoder with four lanes per stream): line schema,
    finite outputs (asserted inside), bpp in a plausible range, both indexed coder tags timed."""
    import argparse
    import json
    import bench
    from sc2bench_amd import hip
    assert 72 > hip.host_coder_max_streams()
    base = dict(workload='mshp224', bs=72, steps=3, warmup=1, no_cpu_baseline=True, coder_group=0, inflight=0, max_inflight=24, ramp=1,
                lag=0, front_priority=0, back_priority=0, coder_priority=0, split_mfma=1, cat_symbols=False, unfused_dequantize=False,
                no_prealloc=False)
    # the module forward per batch (rounds 3 - 4) and the package's stage pipeline (round 5): the same line schema
    bench.workload_bench(argparse.Namespace(no_pipeline=True, **base), dev, 0, 1, False)
    plain = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('{')][-1])
    assert plain['config']['pipeline'].startswith('none')
    bench.workload_bench(argparse.Namespace(no_pipeline=False, **base), dev, 0, 1, False)
    line = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('{')][-1])
    assert line['config']['pipeline']['what'].startswith('sc2bench_amd.pipeline.StagePipeline') and abs(line['bpp'] - plain['bpp']) < 1e-12
    assert line['unit'] == 'images/s' and line['value'] > 0 and line['config']['batch_per_gpu'] == 72
    assert line['config']['range_coder'].startswith('batched HIP coder')
    assert 1.0 < line['bpp'] < 12.0
    assert abs(line['bpp_estimated'] - line['bpp']) < 0.25 * line['bpp']      # the entropy model's estimate of the same batch
    for tag in ('rans_encode.indexed', 'rans_decode.indexed', 'rans_encode', 'rans_decode'):
        assert tag in line['rans'] and line['rans'][tag]['ms_per_launch'] > 0, (tag, line['rans'])
    assert line['roofline'] is not None and 0.0 < line['roofline']['frac'] < 1.0
