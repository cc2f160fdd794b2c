"""`bench.py --workload mshp224 | seg513 | det800x1216 | fp_input` (BASELINE configs 3 - 5 and the hyperprior bottleneck) and the
`secondary` rows of the default line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

from .model import PEAK_BF16_TFLOPS, ROOT, shape_workload, synthetic_batch
from .timing import make_pipeline, ranks_reduced, timed_pipeline_run
from .train import train_bench


def bottleneck_gflop(H, W):
    """algorithmic GFLOP (2 * MACs) of the ten transforms of the FP bottleneck for one H x W image."""
    def o(n, k, st, p):
        return (n + 2 * p - k) // st + 1
    h1, w1 = o(H, 5, 2, 2), o(W, 5, 2, 2)
    h2, w2 = o(h1, 5, 2, 2), o(w1, 5, 2, 2)
    h3, w3 = h2 - 1, w2 - 1
    macs = (h1 * w1 * 96 * (75 + 96) + h2 * w2 * 48 * (2400 + 48) + h3 * w3 * 24 * 192 +
            (h3 + 1) * (w3 + 1) * 512 * (96 + 512) + h3 * w3 * 256 * (2048 + 256) + (h3 + 1) * (w3 + 1) * 256 * 1024)
    return 2e-9 * macs


def _shape_backbone(S, **resnet_kwargs):
    torch.manual_seed(0)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    backbone = S.splittable_resnet(cfg, skips_avgpool=True, skips_fc=True, **resnet_kwargs)
    shape_workload(backbone)
    return backbone


def build_workload(name, dev, bs):
    """The other BASELINE configs as the reference's API runs them (module forward in eval mode after update(): encode ->
    bytes -> decode inside): -> (model, input batch, description, (H, W) of the bottleneck's input or None, default bs)."""
    import sc2bench_amd as S
    from sc2bench_amd import dense, transforms as T
    if name == 'seg513':      # config 5: Entropic-Student DeepLabv3-ResNet-50, PASCAL VOC2012 513 x 513 (voc yaml:132 batch 16)
        n = bs or 16
        backbone = _shape_backbone(S, replace_stride_with_dilation=[False, True, True])
        body = S.FeatureExtractionBackbone(backbone, {'layer3': 'aux', 'layer4': 'out'}, [], False,
                                           analyzable_layer_key='bottleneck_layer')
        model = dense.create_deeplabv3(body, num_input_channels=2048, uses_aux=True, num_aux_channels=1024, num_classes=21)
        model.eval().to(dev)
        model.update()
        body.set_compute_dtype('bf16')
        model.classifier.to(torch.bfloat16)
        model.aux_classifier.to(torch.bfloat16)
        x = torch.rand(n, 3, 513, 513, generator=torch.Generator().manual_seed(0)).to(dev)
        what = ('Entropic-Student DeepLabv3-ResNet-50 (FP bottleneck 24ch, dilated layer3/4 on the HIP head, ASPP head = torch '
                'ops in bf16), 513x513, eval after update()')
        return model, x, what, (513, 513), n
    if name == 'det800x1216':  # config 4: the Faster R-CNN body: bottleneck + FrozenBN layer2-4 + FPN (RPN / RoI heads need torchvision)
        n = bs or 6
        backbone = _shape_backbone(S, norm_layer='FrozenBatchNorm2d')
        model = dense.backbone_with_fpn(backbone, return_layer_dict={'bottleneck_layer': '1', 'layer2': '2', 'layer3': '3', 'layer4': '4'},
                                        in_channels_list=[256, 512, 1024, 2048], out_channels=256,
                                        analyzable_layer_key='bottleneck_layer', analysis_config={'analyzes_after_compress': False})
        model.eval().to(dev)
        model.update()
        model.body.set_compute_dtype('bf16')
        model.fpn.to(torch.bfloat16)
        x = torch.rand(n, 3, 800, 1216, generator=torch.Generator().manual_seed(0)).to(dev)
        what = ('Entropic-Student Faster R-CNN ResNet-50-FPN BODY (FP bottleneck 24ch + FrozenBN layer2-4 on the HIP head + FPN '
                'in bf16 torch ops; RPN / RoI heads need torchvision: not part of this figure), 800x1216, eval after update()')
        return model, x, what, (800, 1216), n
    if name == 'mshp224':      # the mean-scale hyperprior Entropic-Student (29 of the reference's Entropic-Student configs): 224 x 224
        n = bs or 256
        torch.manual_seed(0)
        cfg = {'key': 'MSHPBasedResNetBottleneck', 'kwargs': {'num_latent_channels': 16, 'num_bottleneck_channels': 24,
                                                               'num_target_channels': 256}}
        model = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False, num_classes=1000)
        bl = model.bottleneck_layer
        with torch.no_grad():    # a non-degenerate operating point for random weights: ragged z tables, a latent of std ~1.5,
            eb = bl.entropy_bottleneck      # hyper-synthesis outputs that spread the predicted scales over the scale table
            q = torch.zeros(eb.channels, 1, 3)
            for c in range(eb.channels):
                q[c, 0, 0], q[c, 0, 1], q[c, 0, 2] = -(3 + c % 5), 0.25 * (c % 3), 4 + c % 7
            eb.quantiles.copy_(q)
            bl.g_a[4].weight.mul_(10.0)      # latent std ~1.3
            bl.h_a[2].weight.mul_(4.0)
            w = bl.h_s[4].weight             # [scales | means] halves of gaussian_params (layer.py:764-785 chunks them that way)
            half = w.shape[0] // 2
            w[:half].abs_().mul_(5.0)        # predicted scales ~1.3: the Gaussian model FITS the latent (~2.4 bits per symbol, a
            #                                  few escapes) -- with an untrained h_s every scale sits at the 0.11 floor, every
            #                                  non-zero symbol is bypass-coded and the coder is measured on its slow path only
        model.eval().to(dev)
        model.update()
        model.set_compute_dtype('bf16')
        x = synthetic_batch(n, dev, seed=0)
        what = ('Entropic-Student ResNet-50 with the MEAN-SCALE HYPERPRIOR bottleneck (MSHPBasedResNetBottleneck 16 / 24 ch: g_a, h_a, '
                'h_s, g_s on the HIP kernels; z on the factorised prior, y on the Gaussian conditional with per-symbol CDF rows; both '
                'streams through the batched device coder), 224x224, eval after update(): encode -> bytes -> decode -> layer2..fc')
        return model, x, what, (224, 224), n
    if name == 'fp_input':     # config 3: Factorized-Prior (quality 8) input compression + ResNet-50, 224 x 224
        from sc2bench_amd.resnet import resnet50
        n = bs or 32
        torch.manual_seed(0)
        codec = S.bmshj2018_factorized(8)
        eb = codec.entropy_bottleneck
        with torch.no_grad():
            q = torch.zeros(eb.channels, 1, 3)
            for c in range(eb.channels):
                q[c, 0, 0], q[c, 0, 1], q[c, 0, 2] = -(3 + c % 5), 0.25 * (c % 3), 4 + c % 7
            eb.quantiles.copy_(q)
            codec.g_a[6].weight.mul_(10.0)
        clf = resnet50(num_classes=1000).eval()
        post = T.Compose([T.CenterCrop([224, 224]), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
        model = S.NeuralInputCompressionClassifier(clf, pre_transform=T.AdaptivePad(fill=0, factor=64), compression_model=codec,
                                                   post_transform=post, analysis_config={})
        model.eval().to(dev)
        codec.update()
        model.set_compute_dtype('bf16')      # the ResNet-50 classifier on the library's fused conv + norm kernels (head.HipResNet)
        x = torch.rand(n, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
        what = ('bmshj2018_factorized quality 8 (N 192, M 320) input compression on the HIP kernels (AdaptivePad 64 -> 256x256) + '
                'ResNet-50 classifier (bf16, the library\'s fused conv + norm kernels), 224x224, eval after update()')
        return model, x, what, None, n
    raise SystemExit('unknown workload ' + name)


def workload_bench(args, dev, rank, world, distributed, emit=True, cpu_baseline_fn=None):
    """`--workload mshp224 | seg513 | det800x1216 | fp_input`: that config's updated model through the package's stage pipeline
    (the same scheduler as the headline line: front stages run ahead, the range coder of G batches shares a launch on its own
    HIP stream, byte streams stay on the device), K steps after W warm-up steps.  `--no-pipeline`: the module forward per
    batch (one stream, bytes objects through the host API: the reference's semantics; what rounds 3 - 4 reported).
    `cpu_baseline_fn(workload, model, x)`: bench.py's CPU-baseline leg (the oracle lives there, not here)."""
    from sc2bench_amd import hip
    import sc2bench_amd as S
    model, x, what, hw, n = build_workload(args.workload, dev, args.bs if args.bs != 256 else 0)
    select = lambda tag: tag is not None and (tag.startswith(('enc.', 'dec.', 'g_a', 'g_s', 'h_a', 'h_s')) or tag.startswith('rans'))  # noqa: E731
    pipelined = not args.no_pipeline and S.supports_stages(model)
    pipe = make_pipeline(args, model, dev) if pipelined else None

    def step():
        with torch.no_grad():
            return model(x)

    if pipelined:
        G = pipe.G
        pipe.run(x, n_steps=max(1, (args.warmup + G - 1) // G * G))
        pipe.synchronize()
        if args.warmup > 0 and not args.no_prealloc:
            pipe.warm(x, args.steps)
        if distributed:
            dist.barrier()
        elapsed, _, timer, last, rec = timed_pipeline_run(pipe, x, args.steps, select, distributed)
        out, nb_last, _ = last
        from sc2bench_amd.entropy import _status_or
        assert all(_status_or(st) == 0 for st in rec['statuses']), 'rANS status != 0 in a timed step'
    else:
        for _ in range(max(1, args.warmup)):
            out = step()
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        with hip.KernelTimer(select) as timer:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = step()
            torch.cuda.synchronize(dev)
            if distributed:
                dist.barrier()
            elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    n_ranks = ranks_reduced(dev, distributed)
    leaves = list(out.values()) if isinstance(out, dict) else [out]
    assert all(torch.isfinite(v.float()).all() for v in leaves)
    if rank != 0:
        return None
    ksum = timer.summary()
    bn = {k: v for k, v in ksum.items() if k.startswith(('enc.', 'dec.')) and k != 'dec.dequantize'}
    roofline = None
    if hw is not None and bn:
        ms = sum(v[1] for v in bn.values()) + timer.total_ms('dec.dequantize') / float(args.steps)
        tf = bottleneck_gflop(*hw) * n / ms
        traffic = None   # HBM bytes of the bottleneck forward of one step from the committed PMC passes (tools/pmc_workload.sh)
        tpath = os.path.join(ROOT, 'profiles', 'traffic_workloads.json')
        if os.path.exists(tpath) and args.bs in (0, 256):   # (the committed figures are for the default batch of the workload)
            traffic = json.load(open(tpath)).get(args.workload, {}).get('hbm_bytes_per_bottleneck_forward')
        roofline = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_BF16_TFLOPS,
                    'traffic': traffic, 'kernel': 'bottleneck forward = sum of its fused launches + the dequantise pass', 'kernel_ms': ms,
                    'gflop_per_image': bottleneck_gflop(*hw)}
    # compressed size of the batch as the reference measures it (host API: bytes objects)
    with torch.no_grad():
        bl = model.compression_model if args.workload == 'fp_input' else model.bottleneck_layer if args.workload == 'mshp224' else \
            (model.body if hasattr(model, 'body') else model.backbone).bottleneck_layer
        obj = bl.compress(model.pre_transform(x)) if args.workload == 'fp_input' else bl.encode(x)
    nbytes = sum(len(q) for lst in obj['strings'] for q in lst)     # (the hyperprior codes two streams per image: y and z)
    pix = x.shape[-1] * x.shape[-2] * n
    if pipelined:    # the pipeline's device-resident streams code to the same byte count as the host API's bytes objects
        assert int(nb_last.sum().item()) == nbytes, 'pipeline streams and encode() disagree: {} vs {} bytes'.format(int(nb_last.sum().item()), nbytes)
    # the entropy model's estimate of the same batch: -sum log2 p / pixels in eval mode (sc2bench/loss.py:20-37)
    with torch.no_grad():
        if args.workload == 'fp_input':
            liks = list(model.compression_model(model.pre_transform(x))['likelihoods'].values())
        elif args.workload == 'mshp224':
            bl._forward2train(x)
            liks = list(bl.last_likelihoods)
        else:
            liks = [bl.entropy_bottleneck(bl.analysis(x))[1]]
        bpp_est = float(sum(-torch.log2(v.float()).sum().item() for v in liks)) / pix
    n_streams = len(obj['strings'][0])
    sym_shape = obj.get('shape')
    lat_c = 320 if args.workload == 'fp_input' else 24
    sym_per_stream = lat_c * int(sym_shape[-2]) * int(sym_shape[-1]) if (sym_shape is not None and args.workload != 'mshp224') else \
        '24 x 55 x 55 (y, per-symbol CDF rows) + 16 x {} x {} (z)'.format(int(sym_shape[-2]), int(sym_shape[-1]))
    on_host = (not pipelined) and n_streams <= hip.host_coder_max_streams()
    cpu, cpu_failed = None, None
    if world == 1 and not args.no_cpu_baseline and cpu_baseline_fn is not None:
        try:
            cpu = cpu_baseline_fn(args.workload, model, x)
        except Exception as e:   # the GPU figures are still printed, but a line without its baseline is not a result: rc != 0
            cpu = {'value': None, 'unit': 'images/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: {!r}'.format(e)}
            cpu_failed = 'cpu_baseline failed: {!r}'.format(e)
    if pipelined:
        pl = dict(pipe.describe(), what='sc2bench_amd.pipeline.StagePipeline: front stages run ahead, back stages wait for their coder launch',
                  streams_per_coder_launch=pipe.G * n_streams, coder_group_plan=pipe.group_plan(args.steps)[:6])
        streams = 'device-resident in the timed region (u8 rows in HBM with offset / nbytes vectors)'
        coder = 'batched HIP coder ({} streams of {} symbols per launch)'.format(pipe.G * n_streams, sym_per_stream)
    else:
        pl = 'none: module forward, one stream'
        streams = 'Python bytes through the host API (host coder up to {} streams, batched device coder above)'.format(hip.host_coder_max_streams())
        coder = ('HOST threads (sc2_rans_encode_host / sc2_rans_decode_host): this batch is {} streams of {} symbols, a few long '
                 'serial chains, which a CPU core steps faster than a GPU lane -- these are NOT HIP-coder figures'.format(n_streams, sym_per_stream)) \
            if on_host else 'batched HIP coder ({} streams per launch)'.format(n_streams)
    line = ({
        'metric': 'images/s + bpp, ' + args.workload, 'value': n * args.steps * world / elapsed, 'unit': 'images/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': max(1, args.warmup), 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': what, 'batch_per_gpu': n, 'global_batch': n * world, 'pipeline': pl, 'streams': streams, 'range_coder': coder,
                   'sharding': 'images, no collective', 'ranks_reduced': n_ranks},
        'bpp': 8.0 * nbytes / pix, 'bpp_estimated': bpp_est, 'bytes_per_image': nbytes / n, 'roofline': roofline, 'cpu_baseline': cpu,
        'rans': {k: {'ms_per_launch': round(v[1], 4), 'launches_per_step': v[0] / float(args.steps)}
                 for k, v in sorted(ksum.items()) if k.startswith('rans')},
        'kernels_ms': {k: round(v[1], 4) for k, v in sorted(ksum.items())}})
    if emit:
        print(json.dumps(line))
    if cpu_failed:
        sys.stdout.flush()
        raise SystemExit('bench.py: ' + cpu_failed)
    return line


def secondary_lines(args, dev):
    """Compact rows of the other workloads and of the stage-1 training step, measured by the default invocation after its own
    timed region: {name: {'value', 'unit', 'ms_per_step', 'steps', ...}}; a workload that fails leaves {'error': ...}."""
    import copy
    import gc
    rows = {}
    for name in ('mshp224', 'seg513', 'det800x1216', 'fp_input', 'train_stage1'):
        a = copy.copy(args)
        a.no_cpu_baseline, a.warmup, a.bs, a.coder_group, a.inflight = True, 3, 256, 0, 0
        try:
            if name == 'train_stage1':
                a.mode, a.stage, a.steps, a.warmup = 'train', 1, 10, 3
                line = train_bench(a, dev, 0, 1, False, emit=False)
                rows[name] = {'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'steps': a.steps,
                              'workload': line['config']['workload'], 'batch': line['config']['batch_per_gpu']}
            else:
                a.workload, a.steps = name, 40      # (40 steps, as the stand-alone `--workload` lines: at 20 the coder chains' ramp is a third of the region)
                line = workload_bench(a, dev, 0, 1, False, emit=False)
                rows[name] = {'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'steps': a.steps,
                              'bpp': line['bpp'], 'bpp_estimated': line['bpp_estimated'], 'batch': line['config']['batch_per_gpu'],
                              'pipeline': line['config']['pipeline'] if isinstance(line['config']['pipeline'], str)
                              else {k: line['config']['pipeline'][k] for k in ('steps_per_coder_launch', 'hip_streams', 'streams_per_coder_launch')},
                              'bottleneck_forward_frac_of_mfma_peak': line['roofline']['frac'] if line.get('roofline') else None,
                              'workload': line['config']['workload'][:120]}
        except Exception as e:     # a secondary row never costs the headline line
            rows[name] = {'error': repr(e)[:300]}
        gc.collect()
        torch.cuda.empty_cache()
    return rows
