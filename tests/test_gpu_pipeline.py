"""-m gpu: sc2bench_amd/pipeline.py -- the stage pipeline (front stages ahead, the range coder of G batches in one launch on its
own HIP stream, back stages behind it) produces, batch for batch, exactly what the unpipelined eval forward of the same model
produces (sc2bench/models/backbone.py:229-233 / layer.py:496-521 / 764-817 run one after the other), for every model class that
exposes the three stages: the FP bottleneck classifier, the mean-scale-hyperprior classifier, the DeepLabv3 and FPN bodies and
the neural input-compression classifier; and evaluation.evaluate() on it counts the same hits as the per-batch loop."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _equal(a, b, what):
    if isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()), what
        for k in a:
            _equal(a[k], b[k], '{}[{}]'.format(what, k))
        return
    assert a.shape == b.shape and a.dtype == b.dtype, what
    assert torch.equal(a, b), '{}: pipelined output != unpipelined forward (max diff {})'.format(
        what, (a.float() - b.float()).abs().max().item())


def _run_both(S, model, batches, dev, **pipe_kwargs):
    """every batch through model(x) and through the pipeline; -> (reference outputs, pipelined outputs, nbytes per batch)"""
    assert S.supports_stages(model)
    with torch.no_grad():
        ref = [model(x) for x in batches]
    torch.cuda.synchronize(dev)
    got, nbs = {}, {}

    def keep(step, out, nb, st):
        got[step] = out
        nbs[step] = nb

    pipe = S.StagePipeline(model, dev, **pipe_kwargs)
    rec = {}
    n = pipe.run(list(batches), on_output=keep, record=rec)
    pipe.synchronize()
    assert n == len(batches) and sorted(got) == list(range(len(batches)))
    from sc2bench_amd.entropy import _status_or
    assert all(_status_or(st) == 0 for st in rec['statuses'])
    return ref, [got[i] for i in range(len(batches))], [nbs[i] for i in range(len(batches))]


@pytest.fixture(scope='module')
def bench_mod():
    import bench
    return bench


def test_fp_classifier_pipelined_equals_forward(S, dev, bench_mod):
    """the headline model: five batches, groups of 1 + 2 + 2 (ramp, G = 2), shared symbol buffer; logits bit for bit, and the
    streams' byte counts those of encode()."""
    model = bench_mod.build_model(dev)
    batches = [bench_mod.synthetic_batch(8, dev, seed=s) for s in range(5)]
    ref, got, nbs = _run_both(S, model, batches, dev, coder_group=2, coder_streams=2)
    for i, (a, b) in enumerate(zip(got, ref)):
        _equal(a, b, 'batch {}'.format(i))
    with torch.no_grad():
        enc = model.bottleneck_layer.encode(batches[3])
    assert nbs[3].cpu().tolist() == [len(s) for s in enc['strings'][0]]
    # the same through concatenated payloads (no shared buffer), no ramp, one group of five
    ref2, got2, _ = _run_both(S, model, batches, dev, coder_group=8, coder_streams=1, ramp=False, share_buffer=False)
    for i, (a, b) in enumerate(zip(got2, ref2)):
        _equal(a, b, 'batch {} (cat)'.format(i))


def test_fp_classifier_ragged_last_batch_and_shape_change(S, dev, bench_mod):
    """a loader's last batch is smaller, and a batch of another image size closes its group early: outputs unchanged."""
    model = bench_mod.build_model(dev)
    g = torch.Generator().manual_seed(3)
    batches = [bench_mod.synthetic_batch(6, dev, seed=1), bench_mod.synthetic_batch(6, dev, seed=2),
               (torch.rand(6, 3, 160, 192, generator=g) * 2 - 1).to(dev), bench_mod.synthetic_batch(6, dev, seed=4),
               bench_mod.synthetic_batch(3, dev, seed=5)]
    ref, got, _ = _run_both(S, model, batches, dev, coder_group=4, coder_streams=2, ramp=False)
    for i, (a, b) in enumerate(zip(got, ref)):
        _equal(a, b, 'batch {}'.format(i))


def test_mshp_classifier_pipelined_equals_forward(S, dev, bench_mod):
    """mean-scale hyperprior (layer.py:764-817): y with per-symbol CDF rows rebuilt from the DECODED z, both streams through
    the batched coder, three batches in groups of 1 + 2."""
    model, x, _, _, _ = bench_mod.build_workload('mshp224', dev, 6)
    batches = [bench_mod.synthetic_batch(6, dev, seed=s) for s in (0, 7, 8)]
    ref, got, nbs = _run_both(S, model, batches, dev, coder_group=2, coder_streams=2)
    for i, (a, b) in enumerate(zip(got, ref)):
        _equal(a, b, 'batch {}'.format(i))
    with torch.no_grad():
        enc = model.bottleneck_layer.encode(batches[1])
    assert nbs[1].cpu().tolist() == [len(y) + len(z) for y, z in zip(*enc['strings'])]


@pytest.mark.parametrize('name,n,hw', [('seg513', 2, (513, 513)), ('det800x1216', 1, (320, 416)), ('fp_input', 4, (224, 224))])
def test_dense_and_input_compression_pipelined_equals_forward(S, dev, bench_mod, name, n, hw):
    """DeepLabv3 (dict of resized logits), the FPN body (dict of pyramid levels) and the neural input-compression classifier."""
    model, _, _, _, _ = bench_mod.build_workload(name, dev, n)
    g = torch.Generator().manual_seed(11)
    batches = [torch.rand(n, 3, hw[0], hw[1], generator=g).to(dev) for _ in range(3)]
    if name != 'fp_input':
        # the feature-extraction body (bottleneck + layer2..4 on the library's kernels): bit for bit
        body = model._body()
        ref, got, _ = _run_both(S, body, batches, dev, coder_group=2, coder_streams=2)
        for i, (a, b) in enumerate(zip(got, ref)):
            _equal(a, b, '{} body, batch {}'.format(name, i))
    # the whole model: its head / classifier are torch ops on MIOpen, whose solver choice for a shape may change between the first
    # call and later ones -- equal to a bf16 / f32 rounding step, not necessarily bit for bit
    ref, got, _ = _run_both(S, model, batches, dev, coder_group=2, coder_streams=2)
    for i, (a, b) in enumerate(zip(got, ref)):
        for k in (a.keys() if isinstance(a, dict) else [None]):
            u, v = (a[k], b[k]) if k is not None else (a, b)
            assert u.shape == v.shape and u.dtype == v.dtype
            scale = v.float().abs().max().item()
            assert (u.float() - v.float()).abs().max().item() <= 2.0 ** -6 * scale + 1e-6, '{} batch {} {}'.format(name, i, k)


def test_evaluate_on_the_pipeline_counts_the_same_hits(S, dev, bench_mod):
    """evaluation.evaluate(): pipelined (device-side hit counters, no .item() per batch) == per-batch loop, on a loader whose last
    batch is ragged; a loader of batch size 1 keeps the per-batch forward (the reference's data-size measurement mode)."""
    from sc2bench_amd import evaluation
    model = bench_mod.build_model(dev)
    x = bench_mod.synthetic_batch(22, torch.device('cpu'), seed=2)
    with torch.no_grad():
        labels = model(x.to(dev)).float().argmax(1).cpu()
    labels[::3] = (labels[::3] + 1) % 1000            # two thirds of the top-1 predictions are "right"
    ds = torch.utils.data.TensorDataset(x, labels)
    loader = torch.utils.data.DataLoader(ds, batch_size=8)
    a = evaluation.evaluate(model, loader, dev, pipeline=False)
    b = evaluation.evaluate(model, loader, dev, pipeline_kwargs={'coder_group': 2, 'coder_streams': 2})
    assert b['pipeline'].startswith('stage pipeline') and a['pipeline'].startswith('none')
    assert a['samples'] == b['samples'] == 22
    # (the per-batch loop averages float32 percentages, the pipelined one divides exact hit counts: equal to float32 rounding)
    assert abs(a['acc1'] - b['acc1']) < 1e-4 and abs(a['acc5'] - b['acc5']) < 1e-4 and 50.0 < a['acc1'] < 80.0
    assert abs(b['acc1'] - 100.0 * 14 / 22) < 1e-9
    c = evaluation.evaluate(model, torch.utils.data.DataLoader(ds, batch_size=1), dev, max_samples=4)
    assert c['pipeline'].startswith('none') and c['samples'] == 4


def test_evaluate_pipeline_keeps_each_batch_with_its_own_labels(S, dev, bench_mod):
    """72 batches with DISTINCT label patterns per batch: the caching allocator is in steady state after the first
    few batches, so a label block handed back too early (no record_stream for the back stream that scores it) would be
    overwritten by a later batch's host-to-device copy and batch j scored against other labels.  The per-batch right / wrong
    pattern differs for every batch, so any such mix-up changes the count."""
    from sc2bench_amd import evaluation
    model = bench_mod.build_model(dev)
    n_batches, bs = 72, 2
    x = bench_mod.synthetic_batch(8, torch.device('cpu'), seed=5)
    x = x.repeat(n_batches * bs // 8, 1, 1, 1)
    with torch.no_grad():
        top = torch.cat([model(x[i:i + bs].to(dev)).float().argmax(1).cpu() for i in range(0, len(x), bs)])
    g = torch.Generator().manual_seed(11)
    wrong = torch.rand(len(x), generator=g) < 0.5                      # which images are labelled wrongly: random per image
    labels = torch.where(wrong, (top + 1 + torch.arange(len(x))) % 1000, top)
    expect = 100.0 * float((~wrong).sum()) / len(x)
    ds = torch.utils.data.TensorDataset(x, labels)
    for kw in ({'coder_group': 4, 'coder_streams': 2}, {'coder_group': 8, 'coder_streams': 4, 'back_streams': 2}):
        r = evaluation.evaluate(model, torch.utils.data.DataLoader(ds, batch_size=bs), dev, pipeline=True, pipeline_kwargs=kw)
        assert r['samples'] == len(x) and r['pipeline'].startswith('stage pipeline')
        assert abs(r['acc1'] - expect) < 1e-9, (r['acc1'], expect, kw)


def test_host_coder_steps_give_the_same_outputs(S, dev, bench_mod):
    """StagePipeline(host_steps=k): the first k batches are coded by the host coder on a worker thread, the rest by the device
    coder -- logits, byte counts and statuses equal those of the all-device pipeline bit for bit, in batch order, for a run
    shorter than, equal to and longer than k; auto_host_steps stays within its limits."""
    from sc2bench_amd.pipeline import StagePipeline
    model = bench_mod.build_model(dev)
    xs = [bench_mod.synthetic_batch(16, dev, seed=20 + i) for i in range(7)]

    def run(host_steps, n):
        pipe = StagePipeline(model, dev, coder_group=4, coder_streams=2, host_steps=host_steps)
        outs, rec = [], {}
        pipe.run(iter(xs[:n]), on_output=lambda j, o, nb, st: outs.append((j, o.clone(), nb.clone(), st.clone())), record=rec)
        pipe.synchronize()
        return outs, rec, pipe

    for n in (1, 3, 7):
        ref, _, _ = run(0, n)
        got, rec, pipe = run(3, n)
        assert [j for j, _, _, _ in got] == list(range(n)) and pipe.describe()['host_coder_steps'] == 3
        for (_, a, na, sa), (_, b, nb_, sb) in zip(ref, got):
            assert torch.equal(a, b) and torch.equal(na, nb_) and int(sb.max().item()) == 0 and int(sa.max().item()) == 0
        assert len(rec['statuses']) >= min(n, 3)
    # a tensor input with n_steps (the bench's form) and the warm-up pass
    pipe = StagePipeline(model, dev, coder_group=4, coder_streams=2, host_steps=2)
    pipe.warm(xs[0], 6)
    last = []
    pipe.run(xs[0], n_steps=6, on_output=lambda j, o, nb, st: last.append(o))
    pipe.synchronize()
    assert len(last) == 6 and all(torch.equal(o, last[0]) for o in last)
    assert pipe.group_plan(6, 2) == [1, 1, 4] and pipe.group_plan(20, 3) == [1, 1, 1, 4, 4, 4, 4, 1] and pipe.group_plan(5, 0) == [1, 2, 2]
    assert StagePipeline.auto_host_steps(256, cores=64) in (1, 2, 3, 4) and StagePipeline.auto_host_steps(256, cores=4) == 0
    assert StagePipeline.auto_host_steps(2048, cores=64) == 0


def test_evaluate_with_host_coder_steps_under_inference_mode(S, dev, bench_mod):
    """evaluation.evaluate() runs under torch.inference_mode; the pipeline's host-coder worker thread does not (the mode is per
    thread) and writes the pinned staging buffers: they must not be inference tensors.  Same accuracy as the device-only pipeline."""
    from sc2bench_amd import evaluation
    model = bench_mod.build_model(dev)
    x = bench_mod.synthetic_batch(22, torch.device('cpu'), seed=2)
    with torch.no_grad():
        labels = model(x.to(dev)).float().argmax(1).cpu()
    labels[::3] = (labels[::3] + 1) % 1000
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, labels), batch_size=8)
    a = evaluation.evaluate(model, loader, dev, pipeline_kwargs={'coder_group': 2, 'coder_streams': 2, 'host_steps': 0})
    b = evaluation.evaluate(model, loader, dev, pipeline_kwargs={'coder_group': 2, 'coder_streams': 2, 'host_steps': 2})
    assert a['pipeline'].startswith('stage pipeline') and abs(a['acc1'] - b['acc1']) < 1e-9 and a['samples'] == b['samples'] == 22


def test_pipelines_of_a_process_share_one_pool_of_streams(S, dev, bench_mod):
    """pipeline.pooled_stream: a second StagePipeline of the process runs on the HIP streams the first one ran on (the runtime binds streams
    to hardware queues in the order they are first used -- fresh streams per pipeline made the rows bench.py measures behind its headline run
    depend on how many streams the process had made before), a wider one adds only what it needs beyond them, and two pipelines used one after the
    other on the shared streams still produce the unpipelined forward's outputs."""
    from sc2bench_amd.pipeline import pooled_stream
    model = bench_mod.build_model(dev)
    a = S.StagePipeline(model, dev, coder_group=2, coder_streams=2)
    b = S.StagePipeline(model, dev, coder_group=4, coder_streams=3)
    assert a.front_stream is b.front_stream and a.back_streams[0] is b.back_streams[0] and a.host_out is b.host_out
    assert a.coder_streams[0] is b.coder_streams[0] and a.coder_streams[1] is b.coder_streams[1] and len(b.coder_streams) == 3
    assert pooled_stream(dev, 'coder', 2) is b.coder_streams[2]
    assert pooled_stream(dev, 'coder', 0, priority=-1) is not a.coder_streams[0]       # another priority is another stream
    assert len({s.cuda_stream for s in [b.front_stream, b.host_out] + b.back_streams + b.coder_streams}) == 6
    batches = [bench_mod.synthetic_batch(8, dev, seed=40 + s) for s in range(6)]
    with torch.no_grad():
        ref = [model(x) for x in batches]
    for pipe in (a, b, a):
        got = {}
        pipe.run(list(batches), on_output=lambda step, out, nb, st: got.__setitem__(step, out))
        pipe.synchronize()
        for i in range(len(batches)):
            _equal(got[i], ref[i], 'batch {}'.format(i))
