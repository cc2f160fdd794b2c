"""bench.py -- images/s + bpp of the Entropic-Student ResNet-50 (FP bottleneck 24ch) at 224x224 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bs 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of bs synthetic images resident in HBM, in the
reference's evaluation mode after update() (sc2bench/models/backbone.py:229-233):
    encoder (3 MFMA convs + 2 GDN1) -> symbols -> rANS encode (one stream per image) -> rANS decode ->
    dequantise -> decoder (3 MFMA convs + 2 inverse GDN1) -> ResNet-50 layer2..fc -> logits.
Nothing is skipped: the byte streams are really produced and really decoded; bpp is 8 * bytes / pixels.
Steps run on the PACKAGE's stage pipeline (sc2bench_amd/pipeline.py, `StagePipeline` -- the scheduler
`evaluation.evaluate()` uses on a data loader): the encoder stage on one HIP stream, decoder + head on a second, and the
serial range coder on four coder streams, ONE coder launch per 8 steps (8 x 256 image streams encoded, then decoded, by the
same two serial kernels: their ~20 ms are per-stream latency, not work, and do not grow with the number of streams).
Measured: a long-running kernel on another hardware queue slows every MFMA launch of the pipeline, even a single-thread
spin kernel (6.3 ms per step without the coder, 7.1 ms with spin kernels in its place, 7.9 ms with one coder chain per
step, 6.7 ms with one per 8 steps: DESIGN.md section 6), so fewer, wider coder launches win.  Exactly K steps start and
complete inside the timed region, bracketed by barrier + synchronize, so the region carries one fill and drain of that
pipeline; the wall time is the max over ranks.  One process per GPU; the path shards by image, so N GPUs = N independent
shards, no data-path collective ("weak" scaling, bs per GPU fixed).  `--workload mshp224 | seg513 | det800x1216 |
fp_input` run the other configs' models through the same pipeline class.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (dominant MFMA kernel, HIP events on
its own stream, inside the timed region) and `cpu_baseline` (the oracle = CPU port of the same path, timed on
the host cores of this box on a bounded sample; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# one hardware queue per HIP stream of the software pipeline (the runtime default is 4); must precede HIP init
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
BOTTLENECK_GFLOP_PER_IMG = 8.3418
PEAK_F32_MATRIX_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 (f32 operands): 1/16 of the bf16 rate (MI355X_MICROARCH.md, Matrix cores)
PEAK_HBM_GBS = 8000.0          # HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a streaming copy achieves
# per image, 224x224: algorithmic MFLOP (2 * MACs), bf16 activation MB read, MB written.  Weights
# (< 2.6 MB in total, L2-resident) are not counted.
OPS = {'enc.conv0': (180.6, 0.602, 2.408), 'enc.gdn1': (231.2, 2.408, 2.408), 'enc.conv2': (722.5, 2.408, 0.301),
       'enc.gdn3': (14.5, 0.301, 0.301), 'enc.conv4': (27.9, 0.301, 0.290), 'dec.conv0': (308.3, 0.145, 3.211),
       'dec.igdn1': (1644.2, 3.211, 3.211), 'dec.conv2': (3171.9, 3.211, 1.549), 'dec.igdn3': (396.5, 1.549, 1.549),
       'dec.conv4': (1644.2, 1.549, 1.606),
       # layer2.0's conv1 (256 -> 128) and downsample (256 -> 512, stride 2) when the decoder's last launch takes them along
       # ('dec.conv4+head.2.0'): they read that launch's output tile from LDS and write 56*56*128 + 28*28*512 bf16
       'head.2.0': (411.0, 0.0, 1.606),
       # EntropyModel.dequantize (layer.py:520) = the last pass of the coder's decode launch (rans_dec_finish_dq_kernel): reads
       # the [position][lane] int32 intermediate (72 600 x 4 B), writes the bf16 NHWC latent (72 600 x 2 B); timed by its own
       # event pair (sc2_rans_decode_dequantize_batch_ev); one launch covers every stream of its coder group
       'dec.dequantize': (0.0, 0.2904, 0.1452),
       # round 4's layout pass in front of the first encoder stage (f32 NCHW -> bf16 [N,H,W,4]); since round 5 the first stage
       # reads the f32 planes in place (enc.conv0's 0.602 MB) and this launch only exists with --conv0-layout-pass (A/B)
       'enc.layout': (0.0, 0.602, 0.401)}


def launch_work(tag):
    """(MFLOP, MB) per image of one tagged launch; 'a+b' = ops a and b fused in one launch (reads a's input, writes
    b's output); an unfused GDN launch reads its input twice (GEMM operand + element-wise operand)."""
    parts = [q[:-4] if q.endswith('.f32') else q for q in tag.split('+')]   # '.f32': the reference-precision encoder's launches
    if any(q not in OPS for q in parts):
        return None
    mflop = sum(OPS[q][0] for q in parts)
    rd = OPS[parts[0]][1] * (2 if len(parts) == 1 and 'gdn' in parts[0] else 1)
    return mflop, rd + OPS[parts[-1]][2]


def shape_workload(model):
    """Deterministic, non-degenerate operating point for a random-init model (there are no trained checkpoints
    offline).  With the default init the factorised prior is flat over every table row and the latent rounds to
    {-1, 0, 1}: every image then codes to the same byte count and no escape symbol is ever produced.  Here:
      * quantiles [-(3+c%5), 0.25*(c%3), 4+c%7] per channel c -> ragged tables of 10-19 entries;
      * the first matrix of the cumulative-logit MLP is sharpened per channel (softplus(M0) * 5*(1+0.25*(c%4))): a peaked
        prior, as a trained model has;
      * the last encoder conv is scaled x7: latent std ~1, symbols in about [-6, 6], ~1e-4 escape (bypass) symbols: a
        few per image (0 - 50), as an operating point whose tables fit the latent has (x10 gives 0.7 %).
    Byte counts then depend on the image (synthetic_batch gives every image its own contrast)."""
    import torch.nn.functional as F
    bl = model.bottleneck_layer
    eb = bl.entropy_bottleneck
    with torch.no_grad():
        C = eb.channels
        q = torch.zeros(C, 1, 3)
        k = torch.zeros(C, 1, 1)
        for c in range(C):
            q[c, 0, 0], q[c, 0, 1], q[c, 0, 2] = -(3 + c % 5), 0.25 * (c % 3), 4 + c % 7
            k[c, 0, 0] = 5.0 * (1.0 + 0.25 * (c % 4))
        eb.quantiles.copy_(q.to(eb.quantiles.device))
        m0 = eb.matrices[0]
        m0.copy_(torch.log(torch.expm1(k.to(m0.device) * F.softplus(m0))))
        bl.encoder[4].weight.mul_(7.0)
    return model


def build_model(dev, seed=0, encoder_precision='bf16'):
    import sc2bench_amd as S
    torch.manual_seed(seed)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    model = S.splittable_resnet(cfg, resnet_name='resnet50', skips_avgpool=False, skips_fc=False, num_classes=1000)
    shape_workload(model)
    model.eval().to(dev)
    model.update()
    model.set_compute_dtype('bf16')
    model.set_encoder_precision(encoder_precision)
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)   # the casts above ran on the null stream; the pipeline streams are non-blocking
    return model


def synthetic_batch(bs, dev, seed=0):
    """torch.rand images, each with its own contrast in [0.25, 1] around mid-grey so that the
    compressed size depends on the image, then the ImageNet normalisation of the reference's transform."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.rand(bs, 3, 224, 224, generator=g)
    c = (0.25 + 0.75 * ((torch.arange(bs) * 37) % 64).float() / 63.0).view(bs, 1, 1, 1)
    x = 0.5 + (x - 0.5) * c
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return ((x - mean) / std).to(dev)


def oracle_model(state_dict):
    """The oracle (CPU port) with the device model's parameters; its integer tables are rebuilt by its own update()."""
    from oracle import cpu_ref as R
    ref = R.SplittableResNet50(R.FPBasedResNetBottleneck())
    tables = ('_offset', '_quantized_cdf', '_cdf_length')
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if not k.endswith(tables)}
    ref.load_state_dict({k: v for k, v in sd.items() if not k.startswith('bottleneck_layer.')}, strict=False)
    ref.bottleneck_layer.load_state_dict({k[len('bottleneck_layer.'):]: v for k, v in sd.items()
                                          if k.startswith('bottleneck_layer.')}, strict=False)
    ref.eval()
    ref.update()
    return ref


def oracle_streams(ref, sym_rows, hw):
    """The oracle's range coder (single-threaded C, as upstream) on int32 symbol rows [n, C*hw] -> list[bytes]."""
    from oracle import rans as oracle_rans
    eb = ref.bottleneck_layer.entropy_bottleneck
    n_sym = sym_rows.shape[1]
    idx = (torch.arange(n_sym) // hw).int().numpy()
    cdf, ln, off = eb._quantized_cdf.numpy(), eb._cdf_length.reshape(-1).numpy(), eb._offset.reshape(-1).numpy()
    return [oracle_rans.encode_with_indexes(sym_rows[i].numpy(), idx, cdf, ln, off) for i in range(sym_rows.shape[0])]


def sha256_of(streams):
    import hashlib
    h = hashlib.sha256()
    for s in streams:
        h.update(len(s).to_bytes(4, 'little'))
        h.update(s)
    return h.hexdigest()


def cpu_baseline(sample_images, state_dict, dev_symbols=None, hw=None):
    """The oracle (CPU port of the reference path: torch CPU fp32 ops + single-threaded C rANS, as upstream) on the
    host cores of this box, bounded samples of the same synthetic workload:
      value        full eval path encode -> decode -> head, batches of `sample_images`;
      bs1          the reference's evaluation mode (test batch size 1, yaml:305): encode -> size -> decode -> head per image;
      bs256_train_forward   bottleneck forward in training mode (noise quantisation + likelihoods) in chunks of 32.
    `dev_symbols` (int32 [n, C*hw], the DEVICE's symbols of the first images): the oracle coder's digest of them is
    reported so that the line shows the device bitstreams equal the CPU coder's byte for byte."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 64))   # torch's CPU conv stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(threads)
    ref = oracle_model(state_dict)
    x = synthetic_batch(max(sample_images, 32), torch.device('cpu'), seed=0)
    with torch.no_grad():
        ref(x[:2])  # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            ref(x[:sample_images])
            n += sample_images
            dt = time.perf_counter() - t0
            if dt > 8.0 or n >= 16 * sample_images:
                break
        lens = [len(s) for s in ref.last_encoded['strings'][0]]
        # bs = 1, the reference's evaluation mode
        from oracle import cpu_ref as R
        t1 = time.perf_counter()
        n1, kb = 0, []
        while n1 < 16 and time.perf_counter() - t1 < 5.0:
            ref(x[n1:n1 + 1])
            kb.append(R.file_size(ref.last_encoded))
            n1 += 1
        dt1 = time.perf_counter() - t1
        # bs = 256 bottleneck forward in training mode (likelihood path), chunks of 32
        bl = ref.bottleneck_layer
        ref.train()
        t2 = time.perf_counter()
        n2 = 0
        while n2 < 256 and time.perf_counter() - t2 < 8.0:
            bl._forward2train(x[:32])
            n2 += 32
        dt2 = time.perf_counter() - t2
        ref.eval()
    out = {'value': n / dt, 'unit': 'images/s', 'cores': threads, 'host_cpu_count': os.cpu_count(), 'kind': 'port',
           'sample': '{} images (batches of {}) of the same synthetic workload, full encode->decode->head, '
                     '{:.1f} s of CPU work'.format(n, sample_images, dt),
           'bytes_per_image_min_mean_max': [min(lens), sum(lens) / len(lens), max(lens)],
           'bpp': 8.0 * sum(lens) / (len(lens) * 224 * 224),
           'bs1': {'images_per_s': n1 / dt1, 'ms_per_image': 1e3 * dt1 / n1, 'images': n1, 'data_size_kb_mean': sum(kb) / len(kb),
                   'coder_threads': 1},
           'bs256_train_forward': {'images_per_s': n2 / dt2, 'images': n2, 'chunk': 32,
                                   'what': 'encoder + entropy bottleneck (noise, likelihoods) + decoder, fp32'}}
    if dev_symbols is not None:
        out['bitstream_sha256_first{}'.format(dev_symbols.shape[0])] = sha256_of(oracle_streams(ref, dev_symbols, hw))
    return out


def precision_check(model, x_dev, dev, n=64):
    """Row g3: the SAME n images through (i) the oracle's f32 CPU encoder, (ii) the device's default bf16-MFMA encoder and
    (iii) the device's reference-precision encoder (f32 operands on the f32 matrix cores, set_encoder_precision('f32')):
    symbol mismatch rate, bpp of the streams actually coded from each, and how many images code to the identical bytes.
    Runs after the timed region (the oracle is the checker here, never the thing measured)."""
    ref = oracle_model(model.state_dict())
    configured = model.bottleneck_layer.encoder_precision
    n = min(n, x_dev.shape[0])
    x = x_dev[:n].float().cpu()
    eb, reb = model.bottleneck_layer.entropy_bottleneck, ref.bottleneck_layer.entropy_bottleneck
    pix = x.shape[-1] * x.shape[-2]
    with torch.no_grad():
        ref_lat = torch.cat([ref.bottleneck_layer.encoder(x[i:i + 16]) for i in range(0, n, 16)])
        ref_sym = reb.symbols(ref_lat).reshape(n, -1)
        # estimated rate = what BppLoss trains (sc2bench/loss.py:20-37), in eval mode: -sum log2 p(round(y - m) + m) / pixels
        ref_bpp_est = float(-torch.log2(reb(ref_lat)[1]).sum().item()) / (n * pix)
    hw = None
    out = {'images': n, 'what': 'same images, same weights: f32 CPU oracle encoder vs the device encoders; bpp from the '
                                'streams each one codes (device coder == oracle coder byte for byte given the symbols)'}
    modes = {}
    for mode in ('bf16', 'f32'):
        model.set_encoder_precision(mode)
        with torch.no_grad():
            sym, hw = model.stage_front(x_dev[:n])
            _, _, nb, st = eb.encode_symbols_device(sym, hw[0] * hw[1])
            assert int(st.max().item()) == 0
            # the estimated rate of the same images: eval-mode likelihoods of the f32 latent (eb_forward_kernel)
            lik = eb(model.bottleneck_layer.analysis(x_dev[:n]))[1]
            bpp_est = float(-torch.log2(lik.float()).sum().item()) / (n * pix)
            # encoder-stage time for the whole resident batch
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            model.stage_front(x_dev)
            e0.record()
            for _ in range(3):
                model.stage_front(x_dev)
            e1.record()
            torch.cuda.synchronize(dev)
        s_h = sym.cpu().reshape(n, -1)
        diff = (s_h != ref_sym)
        modes[mode] = {'symbols': s_h, 'nbytes': nb.cpu(),
                       'row': {'symbol_mismatch_rate': diff.float().mean().item(),
                               'images_with_identical_symbols': int((~diff.any(dim=1)).sum().item()),
                               'bpp': 8.0 * float(nb.sum().item()) / (n * pix), 'bpp_estimated': bpp_est,
                               'encoder_stage_ms_per_batch': e0.elapsed_time(e1) / 3.0, 'batch': int(x_dev.shape[0])}}
    model.set_encoder_precision(configured)
    ref_streams = oracle_streams(ref, ref_sym, hw[0] * hw[1])
    ref_len = torch.tensor([len(q) for q in ref_streams])
    out['reference_f32_cpu'] = {'bpp': 8.0 * float(ref_len.sum().item()) / (n * pix), 'bpp_estimated': ref_bpp_est}
    out['bpp_estimated_is'] = ('-sum log2 p(y_hat) / pixels in eval mode, the quantity BppLoss trains (sc2bench/loss.py:20-37); '
                               'bpp = 8 x bytes of the streams actually coded / pixels')
    for mode in ('bf16', 'f32'):
        row = modes[mode]['row']
        row['delta_bpp'] = row['bpp'] - out['reference_f32_cpu']['bpp']
        row['images_with_identical_byte_count'] = int((modes[mode]['nbytes'].long() == ref_len).sum().item())
        out[mode + '_encoder'] = row
    # identical symbols => identical bytes: check it on the f32 encoder's exact images through the device coder
    same = [i for i in range(n) if bool((modes['f32']['symbols'][i] == ref_sym[i]).all())][:8]
    if same:
        with torch.no_grad():
            model.set_encoder_precision('f32')
            sym, hw = model.stage_front(x_dev[:n])
            buf, off, nbs, _ = eb.encode_symbols_device(sym, hw[0] * hw[1])
            model.set_encoder_precision(configured)
            idx = torch.tensor(same, device=dev)
            streams = eb.unpack_strings(buf[idx], off[idx], nbs[idx])
        out['f32_encoder']['bitstreams_identical_to_reference_on_checked_images'] = \
            all(streams[k] == ref_streams[i] for k, i in enumerate(same))
        out['f32_encoder']['images_checked_byte_for_byte'] = len(same)
    return out


STAGE1 = {   # train.stage1 of configs/ilsvrc2012/supervised_compression/entropic_student/splitable_resnet50-fp-beta0.08_from_resnet50.yaml
    'teacher': {'sequential': ['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
                'forward_hook': {'input': [], 'output': ['layer1', 'layer2', 'layer3', 'layer4']}},
    'student': {'sequential': ['bottleneck_layer', 'layer2', 'layer3', 'layer4'],
                'frozen_modules': ['layer2', 'layer3', 'layer4'],
                'forward_hook': {'input': [], 'output': ['bottleneck_layer', 'layer2', 'layer3', 'layer4',
                                                         'bottleneck_layer.entropy_bottleneck']}},
    'optimizer': {'key': 'Adam', 'kwargs': {'lr': 0.001}},
    'criterion': {'key': 'WeightedSumLoss', 'kwargs': {'sub_terms': dict(
        [('layer{}'.format(i), {'criterion': {'key': 'MSELoss', 'kwargs': {'reduction': 'sum'}},
                                'criterion_wrapper': {'key': 'SimpleLossWrapper', 'kwargs': {
                                    'input': {'is_from_teacher': False,
                                              'module_path': 'bottleneck_layer' if i == 1 else 'layer{}'.format(i), 'io': 'output'},
                                    'target': {'is_from_teacher': True, 'module_path': 'layer{}'.format(i), 'io': 'output'}}},
                                'weight': 1.0}) for i in (1, 2, 3, 4)] +
        [('bpp', {'criterion': {'key': 'BppLoss', 'kwargs': {'entropy_module_path': 'bottleneck_layer.entropy_bottleneck',
                                                             'reduction': 'sum'}}, 'weight': 0.08})])}},
}


STAGE2 = {   # train.stage2 of the same YAML (:231-295): KD loss on the logits, decoder + layer2-4 + fc train, encoder + prior frozen
    'teacher': {'sequential': [], 'frozen_modules': [], 'forward_hook': {'input': [], 'output': []}},
    'student': {'sequential': [], 'frozen_modules': ['bottleneck_layer.encoder', 'bottleneck_layer.entropy_bottleneck'],
                'forward_hook': {'input': [], 'output': []}},
    'optimizer': {'key': 'SGD', 'kwargs': {'lr': 0.001, 'momentum': 0.9, 'weight_decay': 0.0005}},
    'criterion': {'key': 'WeightedSumLoss', 'kwargs': {'sub_terms': {'kd': {'criterion': {'key': 'KDLoss', 'kwargs': {
        'student_module_path': '.', 'student_module_io': 'output', 'teacher_module_path': '.', 'teacher_module_io': 'output',
        'temperature': 1.0, 'alpha': 0.5, 'reduction': 'batchmean'}}, 'weight': 1.0}}}},
}


def train_bench(args, dev, rank, world, distributed, emit=True):
    """Stage-1 Entropic-Student training step: frozen teacher forward, student forward (HIP bottleneck + frozen tail),
    MSE-sum + 0.08 * bits, aux loss, backward on the HIP kernels, ONE flat-bucket gradient all-reduce (RCCL), Adam."""
    import sc2bench_amd as S
    from sc2bench_amd import training as T, dataparallel as dp
    from sc2bench_amd.resnet import resnet50
    torch.manual_seed(0)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    student = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False).to(dev)
    teacher = resnet50().to(dev)
    if distributed:
        dp.broadcast_parameters(student)
    stage2 = args.stage == 2
    if stage2:      # the reference updates the bottleneck when stage 2 starts (epoch_to_update): round + detach in the student
        shape_workload(student)
        student.update()
    stage = T.DistillationStage(teacher, student, STAGE2 if stage2 else STAGE1, dev, head_dtype=torch.bfloat16)
    x = synthetic_batch(args.bs, dev, seed=rank)
    targets = torch.randint(0, 1000, (args.bs,), generator=torch.Generator().manual_seed(rank)).to(dev) if stage2 else None

    def step():
        loss = stage.forward_process(x, targets)
        stage.post_forward_process(loss, bottleneck_updated=stage2)
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(loss)
    # metric reduction as evaluation does it (sum of [count, total] over ranks), on the backend's device
    g_images, g_loss = dp.all_reduce_sum_scalars([float(args.bs * args.steps), float(loss) * args.bs])
    line = None
    if rank == 0:
        line = ({
            'metric': 'images/s, Entropic-Student ResNet-50 stage-{} training step, 224^2'.format(args.stage), 'value': args.bs * args.steps * world / elapsed,
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': ('stage 2 of the Entropic-Student recipe (KD loss; decoder + layer2-4 + fc train with batch-statistics '
                                    'BatchNorm: {}; encoder + prior frozen, frozen teacher on the HIP stacks)'.format(
                                        'norm layers + ReLU + residual add on bn.hip, the blocks\' convs on the library\'s kernels under autograd'
                                        if (S.hip.host_policy.bn_train_hip and S.hip.host_policy.conv_train_hip) else
                                        'norm layers on bn.hip, convs on torch / MIOpen' if S.hip.host_policy.bn_train_hip else
                                        'on torch / MIOpen ops under bf16 autocast')) if stage2 else
                                   'stage 1 of the Entropic-Student recipe (bottleneck trains, layer2-4 frozen, frozen teacher)',
                       'batch_per_gpu': args.bs, 'global_batch': args.bs * world, 'gradient_all_reduce_bytes': stage.reducer.nbytes(),
                       'sharding': 'images; one flat-bucket all-reduce per step',
                       'process_group': '{} ({} rank{})'.format(dist.get_backend(), world, '' if world == 1 else 's') if distributed else 'none',
                       'collectives_issued': bool(dp.collectives_active()),
                       'gradient_buckets': len(stage.reducer.buckets),
                       'buckets_launched_from_backward_hooks_last_step': stage.reducer.launched_by_hook,
                       'teacher_on_side_stream': bool(S.hip.host_policy.teacher_stream),
                       'gdn_kernels': 'resident-row (gdn512_rows / gdn96_strips)' if S.hip.host_policy.gdn_rows else 'tile GEMMs',
                       'fused_forward_stages': [n for n, on in (('enc.conv0+gdn96', S.hip.host_policy.train_fused_conv0),
                                                                ('enc.conv2+gdn48', S.hip.host_policy.train_fused_conv2),
                                                                ('dec.conv0+igdn512', S.hip.host_policy.train_fused_dec0)) if on]},
            'final_loss': loss.item(), 'images_all_ranks': g_images, 'mean_loss_all_ranks': g_loss / max(g_images / args.steps, 1.0)})
        if emit:
            print(json.dumps(line))
    if distributed and emit:
        dist.barrier()
        dist.destroy_process_group()
    return line


def dry_run_streams(rank):
    """The per-rank fields of a multi-GPU line (`per_rank`: bpp and the digest of the rank's first 8 byte streams) without a
    device: 8 streams of rank-seeded symbols through the library's HOST range coder (csrc/rans_host.cpp, product code) on the
    known-answer table of tests/golden/rans_kat.json."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from sc2bench_amd import hip
    t = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'rans_kat.json')))['table']
    width = max(len(r) for r in t['cdfs'])
    cdf = np.zeros((len(t['cdfs']), width), np.int32)
    for i, r in enumerate(t['cdfs']):
        cdf[i, :len(r)] = r
    tables = hip.HostRansTables(cdf, t['cdf_sizes'], t['offsets'])
    rng = np.random.RandomState(1000 + rank)
    sym = rng.randint(-3, 4, size=(8, 24 * 55 * 55)).astype(np.int32)
    strings, status = hip.rans_encode_host(tables, sym, index_div=sym.shape[1])      # (every symbol of a stream on table row 0)
    return {'bpp': 8.0 * sum(len(q) for q in strings) / (8 * 224 * 224), 'rans_status': int(status.max()),
            'bitstream_sha256_first8': sha256_of(strings)}


def dry_run(args, world, rank, local_rank):
    """The launch contract without a device: process group (gloo), per-rank shard seed, barrier-bracketed timed region,
    max over ranks, ONE JSON line from rank 0.  No HIP call is made (torch.cuda is not touched)."""
    distributed = world > 1
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))      # stands in for a step; ranks differ so that MAX is exercised
    own_work = time.perf_counter() - t0     # (in front of the closing barrier: what THIS rank took)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # what rank r would process: its own synthetic shard (seed r) = images [lo, hi) of a global batch of bs * world
    sys.path.insert(0, ROOT)
    from sc2bench_amd.dataparallel import shard_range
    lo, hi = shard_range(args.bs * world, rank, world)
    info = {'rank': rank, 'local_rank': local_rank, 'seed': rank, 'shard': [lo, hi], 'own_elapsed_s': elapsed, 'own_work_s': own_work}
    info.update(dry_run_streams(rank))
    ranks = [info]
    n_ranks = None
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        one = torch.ones(1)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)     # backend-side proof of the rank count (the GPU line: `ranks_reduced`)
        n_ranks = int(round(one.item()))
        ranks = [None] * world
        dist.all_gather_object(ranks, info)
    if rank == 0:
        print(json.dumps({'metric': 'images/s + bpp, Entropic-Student ResNet-50 224^2', 'dry_run': True,
                          'value': args.bs * args.steps * world / elapsed, 'unit': 'images/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'none',
                          'config': {'workload': 'dry run: launch / rank / reduction plumbing only',
                                     'batch_per_gpu': args.bs, 'global_batch': args.bs * world, 'ranks_reduced': n_ranks}, 'ranks': ranks}))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


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


def workload_cpu_baseline(name, model, x, budget_s=12.0):
    """The oracle (CPU port, f32 torch CPU ops + the single-threaded C range coder, as upstream) on ONE image of the same
    workload, repeated until `budget_s` seconds of CPU work have run: a reported baseline on a bounded sample (kind 'port').
    seg513 / det800x1216: oracle bottleneck encode -> bytes -> decode, then f32 CPU copies of the model's own tail modules
    (layer2-4 and the ASPP classifier / the FPN); fp_input: the oracle's bmshj2018_factorized + an f32 CPU copy of the classifier."""
    import copy
    from collections import OrderedDict
    from oracle import cpu_ref as R
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 64))
    torch.set_num_threads(threads)
    x1 = x[:1].float().cpu()
    tables = ('_offset', '_quantized_cdf', '_cdf_length')
    with torch.no_grad():
        if name == 'mshp224':
            ref = R.SplittableResNet50(R.MSHPBasedResNetBottleneck())
            sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if not k.endswith(tables + ('scale_table',))}
            ref.load_state_dict({k: v for k, v in sd.items() if not k.startswith('bottleneck_layer.')}, strict=False)
            ref.bottleneck_layer.load_state_dict({k[len('bottleneck_layer.'):]: v for k, v in sd.items()
                                                  if k.startswith('bottleneck_layer.')}, strict=False)
            ref.eval()
            ref.update()

            def run():
                return ref(x1)
            what = 'oracle mean-scale hyperprior bottleneck encode -> bytes -> decode + f32 CPU layer2..fc'
        elif name == 'fp_input':
            from oracle import cpu_ref_input as RI
            codec = model.compression_model
            ref = RI.FactorizedPrior(codec.N, codec.M)
            ref.load_state_dict({k: v.detach().float().cpu() for k, v in codec.state_dict().items() if not k.endswith(tables)}, strict=False)
            ref.eval()
            ref.update(force=True)
            clf = copy.deepcopy(model.classification_model).cpu().float().eval()
            pre, post = model.pre_transform, model.post_transform

            def run():
                return RI.neural_input_compression_forward(pre, ref, post, clf, x1)
            what = 'oracle bmshj2018_factorized compress -> decompress + f32 CPU classifier'
        else:
            body = model.body if hasattr(model, 'body') else model.backbone
            ref_bn = R.FPBasedResNetBottleneck()
            ref_bn.load_state_dict({k: v.detach().float().cpu() for k, v in body.bottleneck_layer.state_dict().items()
                                    if not k.endswith(tables)}, strict=False)
            ref_bn.eval()
            ref_bn.update(force=True)
            tail = [(n_, copy.deepcopy(m).cpu().float().eval()) for n_, m in body.named_children() if n_ != 'bottleneck_layer']
            keys = {str(k): v for k, v in body.return_layer_dict.items()}
            head = copy.deepcopy(model.classifier if name == 'seg513' else model.fpn).cpu().float().eval()

            def run():
                h = ref_bn.decode(**ref_bn.encode(x1))
                feats = OrderedDict()
                if 'bottleneck_layer' in keys:
                    feats[keys['bottleneck_layer']] = h
                for n_, m in tail:
                    h = m(h)
                    if n_ in keys:
                        feats[keys[n_]] = h
                return head(feats['out']) if name == 'seg513' else head(feats)
            what = 'oracle bottleneck encode -> bytes -> decode + f32 CPU copies of layer2-4 and the ' + \
                   ('ASPP classifier' if name == 'seg513' else 'FPN')
        run()   # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            run()
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 8:
                break
    return {'value': n / dt, 'unit': 'images/s', 'cores': threads, 'host_cpu_count': os.cpu_count(), 'kind': 'port',
            'sample': '{} x 1 image of the same workload ({}), {:.1f} s of CPU work'.format(n, what, dt)}


WORKLOAD_PIPELINE = {   # (coder group G, coder streams) per workload, by measurement
    'es224': (8, 4),
    # mean-scale hyperprior: the per-symbol-index decoder holds 120 KB of LDS per workgroup for ~20 ms; 2 048 streams per launch with
    # two 16-stream waves per workgroup (64 CUs held, the library's choice from 1 024 streams up), three launches in flight
    # (tools/mshp_sweep.sh, 40 steps: 35.2 k images/s at G = 2, 36.6 k at G = 8 with one wave per workgroup, 39.3 k with two)
    'mshp224': (8, 3),
    'fp_input': (8, 4),      # 32 streams per batch: 256 per launch
    'seg513': (8, 4),        # 16 streams x 393 k symbols per batch: 128 per launch, ~85 ms of chain each way
    'det800x1216': (8, 6),   # 6 streams x 1.45 M symbols per batch: 48 per launch (one wave), ~330 ms each way
}


def make_pipeline(args, model, dev):
    import sc2bench_amd as S
    g_default, c_default = WORKLOAD_PIPELINE[args.workload]
    return S.StagePipeline(model, dev, coder_group=args.coder_group or g_default, coder_streams=args.inflight or c_default,
                           max_inflight=args.max_inflight, ramp=bool(args.ramp), lag=max(0, args.lag),
                           front_priority=args.front_priority, back_priority=args.back_priority, coder_priority=args.coder_priority,
                           back_streams=max(1, args.split_mfma), share_buffer=not args.cat_symbols,
                           coder_kwargs={'dequantized': False} if args.unfused_dequantize else None)


def timed_pipeline_run(pipe, x, steps, select, distributed, timeline=False):
    """K batches through the package's stage pipeline (sc2bench_amd/pipeline.py), bracketed as the contract says: the caller has
    synchronised; this starts the clock, issues K batches, synchronises every stream (+ barrier) and stops it.
    -> (elapsed s, host issue s, KernelTimer, last (output, nbytes, status), record)"""
    from sc2bench_amd import hip
    rec = {'timeline': []} if timeline else {}
    last = [None]

    def keep(step, out, nb, st):
        last[0] = (out, nb, st)

    with hip.KernelTimer(select) as timer:
        t0 = time.perf_counter()
        pipe.run(x, n_steps=steps, on_output=keep, record=rec)
        t_issued = time.perf_counter()
        pipe.synchronize()
        if distributed:
            dist.barrier()
        t1 = time.perf_counter()
    return t1 - t0, t_issued - t0, timer, last[0], rec


def ranks_reduced(dev, distributed):
    """RCCL-side proof of the rank count: every rank contributes 1 to a device all-reduce on the backend (the process group's
    world size in `config.process_group` comes from the launcher's environment)."""
    if not distributed:
        return None
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(one.item()))


def workload_bench(args, dev, rank, world, distributed, emit=True):
    """`--workload mshp224 | seg513 | det800x1216 | fp_input`: that config's updated model through the package's stage pipeline
    (the same scheduler as the headline line: front stages run ahead, the range coder of G batches shares a launch on its own
    HIP stream, byte streams stay on the device), K steps after W warm-up steps.  `--no-pipeline`: the module forward per
    batch (one stream, bytes objects through the host API: the reference's semantics; what rounds 3 - 4 reported)."""
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
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic_workloads.json')
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
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = workload_cpu_baseline(args.workload, model, x)
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


def self_launch(n):
    """Starts `python -m torch.distributed.run --nproc-per-node n bench.py <the same arguments>` as a child process (one rank per
    GPU over RCCL, rendezvous on 127.0.0.1 and a free port), relays its output and exits with its return code.  A process
    that has initialised the GPU must never be replaced or forked into ranks: this one has not touched it."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print('bench.py: the {}-rank launch failed with return code {}'.format(n, rc), file=sys.stderr)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--inflight', type=int, default=0, help='coder HIP streams (coder launches that may be in flight); 0 = the workload\'s default')
    ap.add_argument('--max-inflight', type=int, default=24, help='encoder stage i waits for decoder+head stage i - this')
    ap.add_argument('--lag', type=int, default=0, help='steps between issuing encoder stage i and decoder+head stage i - lag in host order')
    ap.add_argument('--ramp', type=int, default=1, help='1: the first coder groups of a run hold 1, 2, 4, ... steps')
    ap.add_argument('--coder-group', type=int, default=0, help='steps whose symbols share one range-coder launch; 0 = the workload\'s default')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bs1', action='store_true', help='skip the bs-1 evaluation-mode row')
    ap.add_argument('--no-secondary', action='store_true', help='skip the `secondary` rows (the other workloads + the training step, a few steps each, after the timed region)')
    ap.add_argument('--split-mfma', type=int, default=1, help='decoder+head stages round-robin on K HIP streams of their own')
    ap.add_argument('--front-priority', type=int, default=0, help='HIP stream priority of the encoder stream (-1 = high)')
    ap.add_argument('--back-priority', type=int, default=0, help='HIP stream priority of the decoder+head stream(s) (-1 = high)')
    ap.add_argument('--coder-priority', type=int, default=0, help='HIP stream priority of the coder streams (-1 = high)')
    ap.add_argument('--unfused-dequantize', action='store_true', help='A/B: the coder writes int32 symbols and the decoder+head stage dequantises them (two launches more traffic)')
    ap.add_argument('--cat-symbols', action='store_true', help='A/B: the symbols of a coder group are concatenated (torch.cat) instead of being written into one buffer by the encoder stages')
    ap.add_argument('--conv0-layout-pass', action='store_true', help='A/B: round 4\'s f32 NCHW -> bf16 NHWC4 layout launch in front of the first encoder stage (default: the stage reads the planes in place)')
    ap.add_argument('--no-prealloc', action='store_true', help='A/B: skip the coder-buffer pre-allocation pass after the warm-up steps')
    ap.add_argument('--no-pipeline', action='store_true', help='--workload lines: the module forward per batch (one stream, host bytes) instead of the stage pipeline')
    ap.add_argument('--diag-timeline', action='store_true', help='DIAGNOSTIC: HIP events around every stage of the timed run, printed to stderr (adds ~100 event records)')
    ap.add_argument('--diag-repeat', type=int, default=0, help='DIAGNOSTIC: after the timed region, time R more runs of K steps and print their wall times to stderr')
    ap.add_argument('--policy', default='', help="dispatch-policy overrides, 'field=value,...' (fields of sc2_policy / hip.host_policy): A/B measurements")
    ap.add_argument('--policy-env', action='store_true', help='tools only: also apply the SC2_* variables of the A/B scripts through tools/env_policy.py')
    ap.add_argument('--dry-run', action='store_true', help='rank / shard / barrier / reduction plumbing only (gloo), no GPU call')
    ap.add_argument('--workload', choices=['es224', 'mshp224', 'fp_input', 'seg513', 'det800x1216'], default='es224',
                    help='es224 = the headline config (default); the others are BASELINE configs 3 / 5 / 4 and the hyperprior bottleneck')
    ap.add_argument('--encoder-precision', choices=['bf16', 'f32'], default='bf16',
                    help="f32: the analysis transform with f32 operands on the f32 matrix cores -- symbols, byte streams and bpp are the "
                         "f32 reference path's (precision_check in the line shows it); bf16 (default): the fast encoder")
    ap.add_argument('--stage', type=int, choices=[1, 2], default=1, help='--mode train: which stage of the recipe')
    ap.add_argument('--mode', choices=['infer', 'train'], default='infer',
                    help="'train' = Entropic-Student stage-1 step (secondary figure; the headline metric is 'infer')")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `bench.py --gpus N` on its own: this process becomes the launcher.  It has made no HIP call (importing torch makes
        # none) and makes none: the N ranks are CHILD processes, one per GPU, and this one only relays rank 0's line.
        return self_launch(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.dry_run:
        if args.gpus != world:
            raise SystemExit('bench.py: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
        return dry_run(args, world, rank, local_rank)
    numa = None
    if world > 1:       # (before the first HIP call; one rank alone keeps the whole machine)
        sys.path.insert(0, ROOT)
        from sc2bench_amd.dataparallel import bind_rank_to_gpu_numa
        numa = bind_rank_to_gpu_numa(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # one process per GPU over RCCL.  Under a launcher (torchrun exports RANK / WORLD_SIZE / MASTER_*) the process group is formed
    # whatever the world size, so that `torch.distributed.run --nproc-per-node 1 bench.py` runs the barrier and the MAX
    # reduction of the timing on RCCL on a one-GPU box as well (tests/test_00_rccl_gpu.py); plain `python bench.py` forms none.
    launched = all(k in os.environ for k in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'))
    distributed = world > 1 or launched
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE {}: the line would not describe the run that was asked for'
                         .format(args.gpus, world))

    import sc2bench_amd as S
    from sc2bench_amd import hip
    from sc2bench_amd.entropy import _status_or
    if args.policy or args.policy_env:      # A/B runs: the package itself reads no SC2_* dispatch variable
        from tools import env_policy
        overrides = env_policy.apply() if args.policy_env else {}
        overrides.update(env_policy.parse(args.policy))
        if overrides:
            hip.configure(**overrides)
        print('bench.py: dispatch policy overrides {}'.format(overrides), file=sys.stderr)
    if args.mode == 'train':
        return train_bench(args, dev, rank, world, distributed)
    if args.workload != 'es224':
        workload_bench(args, dev, rank, world, distributed)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    model = build_model(dev, encoder_precision=args.encoder_precision)
    if args.conv0_layout_pass:
        model.bottleneck_layer.conv0_reads_nchw = False
    x = synthetic_batch(args.bs, dev, seed=rank)   # a different shard per rank, resident in HBM
    torch.cuda.synchronize(dev)
    # The software pipeline is the package's (sc2bench_amd/pipeline.py: StagePipeline over stage_front / stage_coder /
    # stage_back of the model): front(i) [encoder + quantise] stages run ahead on one HIP stream, the serial range coder
    # (encode -> bytes -> decode) of up to G consecutive steps runs as ONE launch on one of the coder streams, back(i)
    # [dequantise + decoder + head] waits for its coder launch on a second MFMA stream.  evaluation.evaluate() runs the same
    # class on a data loader; bench.py only feeds it the resident synthetic batch K times and reads the clock.
    pipe = make_pipeline(args, model, dev)
    G, n_coder = pipe.G, len(pipe.coder_streams)
    with torch.no_grad():
        # fold / pack every cached weight once on the null stream, before the side streams use them, THROUGH THE THREE STAGES the
        # pipeline runs (the staged form packs more than forward_device does: the symbol-writing last encoder conv and the
        # decoder tail that carries layer2.0's two 1x1 layers)
        sym0, hw0 = model.stage_front(x[:2])
        dec0, _, st0 = model.stage_coder(sym0, hw0, **pipe.coder_kwargs)
        model.stage_back(dec0, hw0)
        assert int(st0.max().item()) == 0
        del sym0, dec0, st0
    torch.cuda.synchronize(dev)

    def sync_all():
        pipe.synchronize()
        if distributed:
            dist.barrier()

    # whole coder groups, so that the timed region meets warm allocator pools and LDS attributes (W = 0 stays 0)
    warm_steps = (args.warmup + G - 1) // G * G
    if warm_steps:
        pipe.run(x, n_steps=warm_steps)
    sync_all()
    if args.warmup > 0 and not args.no_prealloc:
        # resource warm-up, not a step: one untimed range-coder launch per coder-group shape of the timed plan, so that the timed
        # region makes no first-time device allocation (the FIRST process on a freshly booted box pays ~45 ms for them otherwise)
        pipe.warm(x, args.steps)
        sync_all()

    select = lambda tag: launch_work(tag) is not None or tag.startswith('rans')  # noqa: E731
    elapsed, issue_s, timer, last, rec = timed_pipeline_run(pipe, x, args.steps, select, distributed, timeline=args.diag_timeline)
    statuses, latency = rec['statuses'], rec['latency']
    if args.diag_timeline:
        tl = rec['timeline']
        base = tl[0][2]
        for kind, step, e_a, e_b in sorted(tl, key=lambda r: base.elapsed_time(r[2])):
            print('timeline {:<6} step {:3d}  start {:8.2f} ms  end {:8.2f} ms  ({:.2f} ms)'.format(
                kind, step, base.elapsed_time(e_a), base.elapsed_time(e_b), e_a.elapsed_time(e_b)), file=sys.stderr)
    for rep in range(args.diag_repeat):
        tr0 = time.perf_counter()
        pipe.run(x, n_steps=args.steps)
        sync_all()
        print('diag-repeat {}: first timed region {:.2f} ms, this one {:.2f} ms'.format(rep, elapsed * 1e3, (time.perf_counter() - tr0) * 1e3),
              file=sys.stderr)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    n_ranks = ranks_reduced(dev, distributed)

    logits, nb, st = last
    assert all(_status_or(s_) == 0 for s_ in statuses), 'rANS status != 0 in a timed step'
    assert torch.isfinite(logits.float()).all()
    nb_f = nb.float()
    bytes_per_img = nb_f.mean().item()
    bpp = 8.0 * bytes_per_img / (224 * 224)
    images = args.bs * args.steps * world
    value = images / elapsed

    per_rank = None
    if distributed:
        # every rank's own bpp and the digest of the byte streams of the first 8 images of ITS shard travel to rank 0 as device
        # tensors on the backend (RCCL cannot gather host objects): a multi-GPU line can be judged for bitstream / bpp parity
        # rank by rank, not only through rank 0's shard
        with torch.no_grad():
            sym_r, hw_r = model.stage_front(x)
            eb_r = model.bottleneck_layer.entropy_bottleneck
            buf_r, off_r, nbs_r, sts_r = eb_r.encode_symbols_device(sym_r[:8], hw_r[0] * hw_r[1])
            digest = sha256_of(eb_r.unpack_strings(buf_r, off_r, nbs_r))
            del sym_r, buf_r
        mine = torch.cat([torch.tensor(list(bytes.fromhex(digest)), dtype=torch.float64, device=dev),
                          torch.tensor([bpp, float(rank), float(int(sts_r.max().item()))], dtype=torch.float64, device=dev)])
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = [{'rank': int(r[33].item()), 'bpp': r[32].item(), 'rans_status': int(r[34].item()),
                     'bitstream_sha256_first8': bytes(int(v) for v in r[:32].tolist()).hex()} for r in rows]

    if rank == 0:
        ksum = timer.summary()
        # every launch of the bottleneck forward: the fused conv / GDN launches (mean duration per launch = per step) and the
        # coder's dequantise pass, whose launches cover 1 .. G steps each: its total over the run / steps
        conv = {k: v for k, v in ksum.items() if launch_work(k) is not None and k != 'dec.dequantize'}
        dq_ms = timer.total_ms('dec.dequantize') / float(args.steps)
        dom = max(conv, key=lambda k: conv[k][0] * conv[k][1])
        fwd_ms = sum(v[1] for v in conv.values()) + dq_ms
        # (the last decoder launch may carry two 1x1 layers of the head: their 0.411 GFLOP per image then sit in fwd_ms too)
        fwd_gflop = BOTTLENECK_GFLOP_PER_IMG + (0.411 if any('head.2.0' in k for k in conv) else 0.0)

        def roof(k, ms):
            """bound = whichever roof the launch's algorithmic intensity puts it under (ridge = 312.5 FLOP/B)"""
            mflop, mbyte = launch_work(k)
            f32 = k.endswith('.f32')     # the reference-precision encoder: f32 activations (twice the bytes), f32 matrix peak
            mbyte = mbyte * (2.0 if f32 else 1.0)
            peak_tf = PEAK_F32_MATRIX_TFLOPS if f32 else PEAK_BF16_TFLOPS
            tf = mflop * 1e6 * args.bs / (ms * 1e-3) / 1e12
            gbs = mbyte * 1e6 * args.bs / (ms * 1e-3) / 1e9
            hbm = mflop / mbyte < peak_tf * 1e3 / PEAK_HBM_GBS
            return {'bound': 'hbm' if hbm else 'mfma', 'achieved': gbs if hbm else tf,
                    'peak': PEAK_HBM_GBS if hbm else peak_tf, 'unit': 'GB/s' if hbm else 'TFLOP/s',
                    'frac': (gbs / PEAK_HBM_GBS) if hbm else (tf / peak_tf), 'tflops': tf, 'gbs': gbs,
                    'kernel_ms': ms}
        per_kernel = {k: roof(k, v[1]) for k, v in conv.items()}
        if dq_ms > 0:
            per_kernel['dec.dequantize'] = roof('dec.dequantize', dq_ms)
        floor_ms = sum(max(launch_work(k)[0] * 1e6 * args.bs / (PEAK_BF16_TFLOPS * 1e12),
                           launch_work(k)[1] * 1e6 * args.bs / (PEAK_HBM_GBS * 1e9)) * 1e3 for k in per_kernel)
        # HBM-side bytes per launch of the dominant kernel from the committed PMC pass -- reported only while that pass still
        # describes the library that runs (profiles/traffic.json records the hashes of the library / kernel sources it measured)
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                meas, now = tj.get('_measured_on') or {}, hip.library_fingerprint()
                same = any(meas.get(k) and meas.get(k) == now.get(k) for k in ('lib_sha256', 'csrc_sha256'))
                rec_t = tj.get(dom)
                if rec_t and same:      # measured at bs 256 per GPU; scaled to this run's batch
                    traffic = rec_t['hbm_bytes_per_launch'] * args.bs / 256.0
                elif rec_t:
                    traffic_note = 'profiles/traffic.json was measured on another build of the library: dropped'
            except Exception:
                traffic = None
        lat_ms = [e0.elapsed_time(e1) for _, e0, e1 in latency if e1 is not None]
        # the same launches stand-alone: one stream, nothing beside them (the in-pipeline durations above are taken while the
        # other MFMA stream and the coder share the chip with the launch: the sum of overlapped kernel durations counts shared
        # time twice)
        with torch.no_grad():
            sym_s, hw_s = model.stage_front(x)
            dec_s, _, _ = model.stage_coder(sym_s, hw_s, **pipe.coder_kwargs)
            torch.cuda.synchronize(dev)
            with hip.KernelTimer(select) as solo:
                for _ in range(5):
                    model.stage_front(x)
                    model.stage_back(dec_s, hw_s)
                torch.cuda.synchronize(dev)
            solo_sum = {k: v for k, v in solo.summary().items() if launch_work(k) is not None}
            solo_ms = sum(v[1] for v in solo_sum.values())
            # ... and on SURVEY 8(d)'s own basis: the ten layers of the bottleneck and nothing else (8.3418 GFLOP per image), i.e.
            # with the decoder's last conv as a launch of its own instead of the form that carries layer2.0's two 1x1 layers
            with hip.KernelTimer(select) as solo_p:
                for _ in range(5):
                    model.stage_front(x)
                    model.stage_decoder(dec_s, hw_s)
                torch.cuda.synchronize(dev)
            plain_sum = {k: v for k, v in solo_p.summary().items() if launch_work(k) is not None}
            plain_ms = sum(v[1] for v in plain_sum.values())
            del sym_s, dec_s
        # steady state of the dominant kernel: a tail run of >= 60 steps after the timed region, the first 8
        # launches dropped -- with the pipeline full the encoder stage of later steps and the coder share the CUs with it, which a
        # 20-step region barely reaches
        steady_steps = max(60, args.steps)
        with hip.KernelTimer(lambda tag: tag == dom) as steady_t:
            pipe.run(x, n_steps=steady_steps)
            sync_all()
        steady = [r[1].elapsed_time(r[2]) for r in steady_t.records if r[0] == dom][8:]
        steady_ms = sum(steady) / max(1, len(steady))
        # the device bitstreams of the first 8 images of this shard, and the symbols they were coded from
        with torch.no_grad():
            sym, hw = model.stage_front(x)
            eb = model.bottleneck_layer.entropy_bottleneck
            buf, off, nbs, sts = eb.encode_symbols_device(sym, hw[0] * hw[1])
            assert int(sts.max().item()) == 0
            dev_streams = eb.unpack_strings(buf[:8], off[:8], nbs[:8])
            sym8 = sym[:8].cpu()
        out = {
            'metric': 'images/s + bpp, Entropic-Student ResNet-50 224^2',
            'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'Entropic-Student ResNet-50 (FPBasedResNetBottleneck 24ch), ILSVRC2012 shape '
                                   '224x224x3, eval after update(): encode -> rANS -> decode -> layer2..fc',
                       'batch_per_gpu': args.bs, 'global_batch': args.bs * world,
                       'pipeline': 'sc2bench_amd.pipeline.StagePipeline (the scheduler evaluation.evaluate() uses): encoder stages run '
                                   'ahead, decoder+head stages wait for their coder launch',
                       'hip_streams': pipe.describe()['hip_streams'],
                       'steps_per_coder_launch': G, 'max_inflight_steps': args.max_inflight, 'coder_group_plan': pipe.group_plan(args.steps)[:6], 'warmup_steps_run': warm_steps,
                       'prealloc': 'none' if (args.no_prealloc or args.warmup == 0) else 'one untimed range-coder launch per coder group of the timed plan (device buffers only, not a step)',
                       'weights': 'random init seed 0, operating point shaped by bench.shape_workload (ragged tables, '
                                  'peaked prior, latent std ~1, ~1e-4 escape symbols)',
                       'images': 'torch.rand, per-image contrast 0.25-1, ImageNet normalisation',
                       'streams': 'device-resident in the timed region (u8 rows in HBM with offset / nbytes vectors, no Python '
                                  'bytes objects); the host-bytes encode()/decode() API of the reference is timed in bs1_eval',
                       'encoder_precision': args.encoder_precision + (' (f32 operands on v_mfma_f32_16x16x4_f32: bitstreams of the f32 '
                                                                      'reference path; decoder + head bf16)' if args.encoder_precision == 'f32' else
                                                                      ' MFMA operands, f32 accumulation'),
                       'first_encoder_stage_input': 'round-4 layout pass (A/B)' if args.conv0_layout_pass else 'f32 NCHW planes read in place',
                       'sharding': 'images, no collective',
                       'process_group': '{} ({} rank{})'.format(dist.get_backend(), world, '' if world == 1 else 's') if distributed else 'none',
                       'ranks_reduced': n_ranks,
                       'numa_binding': numa if world > 1 else 'none (one rank)'},
            'bpp': bpp, 'bytes_per_image': bytes_per_img,
            'bytes_per_image_min_mean_max': [nb_f.min().item(), bytes_per_img, nb_f.max().item()],
            'bitstream_sha256_first8': sha256_of(dev_streams),
            'latency_ms_per_batch': {'mean': sum(lat_ms) / max(1, len(lat_ms)), 'max': max(lat_ms) if lat_ms else None,
                                     'what': 'encoder stage start -> logits of the same batch'},
            'host_issue_ms_per_step': 1e3 * issue_s / args.steps,
            'roofline': dict(per_kernel[dom], kernel=dom, traffic=traffic, launches_timed=conv[dom][0],
                             frac_steady=roof(dom, steady_ms)['frac'],
                             steady={'kernel_ms': steady_ms, 'launches_averaged': len(steady), 'launches_dropped': 8,
                                     'what': 'the same kernel in a tail run of {} steps of the same pipeline after the timed region '
                                             '(pipeline full: the encoder stages of later steps and the coder share the CUs)'.format(steady_steps)}),
            'bottleneck_forward': {'ms_per_batch_sum_of_its_launches': fwd_ms,
                                   'launches_counted': sorted(per_kernel),
                                   'gflop_per_image': fwd_gflop,
                                   'tflops': fwd_gflop * args.bs / fwd_ms,
                                   'frac_of_mfma_peak': fwd_gflop * args.bs / fwd_ms / PEAK_BF16_TFLOPS,
                                   # SURVEY 8(d)'s numerator only: the rider's FLOPs left out, its time left IN (a lower bound)
                                   'frac_8p3418': BOTTLENECK_GFLOP_PER_IMG * args.bs / fwd_ms / PEAK_BF16_TFLOPS,
                                   'head_2_0_rider': {'gflop_per_image': fwd_gflop - BOTTLENECK_GFLOP_PER_IMG,
                                                      'stand_alone_ms': solo_ms - plain_ms,
                                                      'what': 'layer2.0 conv1 + downsample of the task head, carried by the decoder\'s last launch '
                                                              '(dec.conv4+head.2.0); stand_alone_ms = that launch minus dec.conv4 as a launch of its own'},
                                   'roofline_floor_ms_of_this_launch_structure': floor_ms,
                                   'frac_of_floor': floor_ms / fwd_ms,
                                   'stand_alone': {'ms_per_batch': solo_ms, 'frac_of_mfma_peak': fwd_gflop * args.bs / solo_ms / PEAK_BF16_TFLOPS,
                                                   'what': 'the same launches (without the dequantise pass) on one stream with nothing beside them, 5 repetitions after the timed region',
                                                   'kernels_ms': {k: round(v[1], 4) for k, v in sorted(solo_sum.items())},
                                                   'frac_8p3418': BOTTLENECK_GFLOP_PER_IMG * args.bs / plain_ms / PEAK_BF16_TFLOPS,
                                                   'ms_per_batch_8p3418': plain_ms,
                                                   'kernels_ms_8p3418': {k: round(v[1], 4) for k, v in sorted(plain_sum.items())},
                                                   'what_8p3418': 'SURVEY 8(d) basis: the bottleneck\'s ten layers alone (dec.conv4 as a launch of its '
                                                                  'own, no head layer inside), 8.3418 GFLOP per image'}},
            'kernel_rooflines': {k: {'bound': v['bound'], 'frac': round(v['frac'], 4), 'tflops': round(v['tflops'], 1),
                                     'gbs': round(v['gbs'], 1), 'ms': round(v['kernel_ms'], 4)}
                                 for k, v in sorted(per_kernel.items())},
            'kernels_ms': {k: round(v[1], 4) for k, v in sorted(ksum.items())},
        }
        if traffic_note:
            out['roofline']['traffic_note'] = traffic_note
        if per_rank is not None:
            out['per_rank'] = sorted(per_rank, key=lambda r: r['rank'])
            out['config']['ranks_gathered'] = len(per_rank)
        if 'rans_encode' in ksum:
            n_sym = 24 * 55 * 55
            out['rans'] = {'encode_ms': ksum['rans_encode'][1], 'decode_ms': ksum['rans_decode'][1],
                           'streams_per_launch': G * args.bs, 'launches_in_flight': n_coder,
                           'streams_in_flight': G * args.bs * n_coder, 'symbols_per_stream': n_sym,
                           'encode_Msym_per_s_per_stream': n_sym / ksum['rans_encode'][1] / 1e3,
                           'decode_Msym_per_s_per_stream': n_sym / ksum['rans_decode'][1] / 1e3}
        failed = None
        if world == 1 and not args.no_cpu_baseline:
            out['precision_check'] = precision_check(model, x, dev)
            out['symbol_mismatch_rate'] = out['precision_check'][args.encoder_precision + '_encoder']['symbol_mismatch_rate']
            out['delta_bpp'] = out['precision_check'][args.encoder_precision + '_encoder']['delta_bpp']
            out['bpp_estimated'] = out['precision_check'][args.encoder_precision + '_encoder']['bpp_estimated']
        if world == 1 and not args.no_cpu_baseline and args.encoder_precision == 'bf16':
            # the same K steps of the same pipeline with the reference-precision encoder (f32 operands on the f32 matrix cores):
            # the mode whose bitstreams are the f32 reference path's, measured by the same command as the headline figure
            model.set_encoder_precision('f32')
            if warm_steps:
                pipe.run(x, n_steps=warm_steps)
            sync_all()
            el32, _, _, last32, rec32 = timed_pipeline_run(pipe, x, args.steps, lambda tag: False, distributed)
            model.set_encoder_precision('bf16')
            _, nb32, _ = last32
            assert all(_status_or(s_) == 0 for s_ in rec32['statuses']), 'rANS status != 0 in an f32-mode step'
            pc32 = out['precision_check']['f32_encoder']
            out['f32_mode'] = {'images_per_s': args.bs * args.steps / el32, 'ms_per_step': 1e3 * el32 / args.steps,
                               'steps': args.steps, 'bpp': 8.0 * nb32.float().mean().item() / (224 * 224),
                               'bpp_estimated': pc32['bpp_estimated'], 'symbol_mismatch_rate': pc32['symbol_mismatch_rate'],
                               'images_with_identical_symbols': '{} of {}'.format(pc32['images_with_identical_symbols'],
                                                                                  out['precision_check']['images']),
                               'what': 'the same pipeline with set_encoder_precision("f32"): symbols / bitstreams of the f32 reference '
                                       'path (the residual mismatch is f32 summation order against torch CPU); decoder + head bf16'}
        if world == 1 and not args.no_bs1:
            out['bs1_eval'] = bs1_eval(model, x, dev)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(8, model.state_dict(), dev_symbols=sym8, hw=hw[0] * hw[1])
                out['bitstream_match'] = out['cpu_baseline'].get('bitstream_sha256_first8') == out['bitstream_sha256_first8']
            except Exception as e:  # the GPU figures are still printed, but a line without its baseline is not a result: rc != 0
                out['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': 'failed: {!r}'.format(e)}
                failed = 'cpu_baseline failed: {!r}'.format(e)
        if world == 1 and not distributed and not args.no_secondary:
            # the other configurations through the SAME command (VERDICT r4 "builder-only numbers"): after the headline's timed
            # region, each on the package's stage pipeline (or the training step), a few steps each, compact rows in `secondary`
            del logits, nb, st, last
            out['secondary'] = secondary_lines(args, dev)
        print(json.dumps(out))
        if failed:
            sys.stdout.flush()
            raise SystemExit('bench.py: ' + failed)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def bs1_eval(model, x, dev, n=64):
    """The reference's evaluation mode (script/task/image_classification.py:106-145, test batch size 1): per image
    forward() = encode -> FileSizeAnalyzer on the pickled {'strings','shape'} -> decode -> head, through the host API
    (bytes objects cross to the host and back, as in the reference).  Round 6: the two device halves of that forward replay HIP
    graphs (sc2bench_amd/graphs.py) around the host range coder; the row also carries the eager figure, the launch count of an
    eager forward and per-image latency percentiles (one synchronize per image)."""
    import sc2bench_amd as S
    from sc2bench_amd import hip
    model.analyzes_after_compress = True
    model.analyzers = [S.FileSizeAnalyzer(unit='KB')]
    model.activate_analysis()

    def loop(count):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(count):
            model(x[i % x.shape[0]:i % x.shape[0] + 1])
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    with torch.no_grad():
        graphs_on = bool(hip.host_policy.eval_graphs)
        hip.configure(eval_graphs=False)
        for i in range(3):
            model(x[i:i + 1])
        with hip.KernelTimer() as kt:           # every tagged launch of ONE eager forward
            model(x[0:1])
            torch.cuda.synchronize(dev)
        launches = len(kt.records)
        dt_eager = loop(n)
        hip.configure(eval_graphs=graphs_on)
        for i in range(3):
            model(x[i:i + 1])
        used = model.__dict__.get('_eval_graphs') is not None and any(isinstance(v, S.graphs.EvalGraphs) for v in model.__dict__['_eval_graphs'].values())
        model.clear_analysis()
        dt = loop(n)
        sizes = model.analyzers[0].file_size_list[-n:]
        lat = []
        for i in range(n):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            model(x[i % x.shape[0]:i % x.shape[0] + 1])
            torch.cuda.synchronize(dev)
            lat.append(1e3 * (time.perf_counter() - t0))
        lat.sort()
    model.deactivate_analysis()
    return {'images_per_s': n / dt, 'ms_per_image': 1e3 * dt / n, 'images': n,
            'data_size_kb_mean': sum(sizes) / len(sizes),
            'latency_ms': {'p50': lat[len(lat) // 2], 'p99': lat[min(len(lat) - 1, int(0.99 * len(lat)))], 'what': 'one synchronize per image'},
            'hip_graphs': ('2 graph replays per image (encoder | dequantise + decoder + layer2..fc) around the host range coder, '
                           'captured once per input shape in this process' if used else
                           'off: ' + str(model.__dict__.get('_eval_graphs_error') or 'policy')),
            'eager': {'ms_per_image': 1e3 * dt_eager / n, 'launches_per_image': launches},
            'what': 'bs 1, forward() with host bytes (encode -> pickle size -> decode -> layer2..fc), one stream'}


if __name__ == '__main__':
    main()
