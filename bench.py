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
serial range coder on three coder streams, ONE coder launch per 8 steps (8 x 256 image streams encoded, then decoded, by the
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
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')   # encoder, decoder + head, 3 coder streams, the host-coder stream, the null stream, spare

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Round 6: the harness is split by concern (benchlib/); this file keeps the argument parser, the headline line and the
# CPU-baseline leg -- the only code outside tests/ and __graft_entry__.smoke() that touches oracle/.  The names below are
# re-exported: the tests and tools/ import them from `bench`.
from benchlib.model import (BOTTLENECK_GFLOP_PER_IMG, OPS, PEAK_BF16_TFLOPS, PEAK_F32_MATRIX_TFLOPS, PEAK_HBM_GBS,  # noqa: E402,F401
                            build_model, launch_work, sha256_of, shape_workload, synthetic_batch)
from benchlib.timing import WORKLOAD_PIPELINE, make_pipeline, ranks_reduced, timed_pipeline_run  # noqa: E402,F401
from benchlib.train import STAGE1, STAGE2, train_bench  # noqa: E402,F401
from benchlib.workloads import bottleneck_gflop, build_workload, secondary_lines, workload_bench  # noqa: E402,F401
from benchlib.launch import dry_run, dry_run_streams, self_launch  # noqa: E402,F401
from benchlib.eval_mode import bs1_eval  # noqa: E402,F401


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
    ap.add_argument('--host-steps', type=int, default=-1, help='leading steps whose streams the HOST coder codes (sc2bench_amd/pipeline.py); -1 = auto from the core count, 0 = none')
    ap.add_argument('--host-ramp-skip', type=int, default=1, help='A/B: 1 = behind host-coded steps the device coder ramp starts at 2^host_steps steps per launch; 0 = at 1')
    ap.add_argument('--numa-bind', choices=['auto', 'on', 'off'], default='auto', help="bind this process to the CPUs of its GPU's NUMA node before the first HIP call (auto: when there is more than one rank)")
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
    if args.numa_bind == 'on' or (args.numa_bind == 'auto' and world > 1):       # (before the first HIP call; by default one rank alone keeps the whole machine)
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
        workload_bench(args, dev, rank, world, distributed, cpu_baseline_fn=workload_cpu_baseline)
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
    if args.diag_timeline:
        pipe._host_staging['_trace'] = []
    elapsed, issue_s, timer, last, rec = timed_pipeline_run(pipe, x, args.steps, select, distributed, timeline=args.diag_timeline)
    statuses, latency = rec['statuses'], rec['latency']
    if args.diag_timeline:
        for tr in pipe._host_staging.get('_trace', []):
            print('host coder job {}'.format(tr), file=sys.stderr)
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
                       'steps_per_coder_launch': G, 'max_inflight_steps': args.max_inflight, 'coder_group_plan': pipe.group_plan(args.steps, pipe.resolve_host_steps(x))[:8],
                       'host_coder_steps': {'steps': pipe.resolve_host_steps(x), 'host_cores': hip.host_cores(), 'host_ms_per_batch_measured': pipe.__dict__.get('_host_ms', {}).get(int(x.shape[0])),
                                            'what': 'the first batches of the run are rANS-coded (encode + decode, the same bytes) by the library\'s host coder on the CPU cores while the device coder\'s first group is under way: sc2bench_amd/pipeline.py'}, 'warmup_steps_run': warm_steps,
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
                       'numa_binding': numa if (world > 1 or args.numa_bind == 'on') else 'none (one rank)'},
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
                                   # ... and with the rider's time taken out too, by its stand-alone cost (an estimate: inside the
                                   # pipeline the launch that carries it stretches like every other)
                                   'frac_8p3418_rider_time_removed': BOTTLENECK_GFLOP_PER_IMG * args.bs / max(1e-9, fwd_ms - max(0.0, solo_ms - plain_ms)) / PEAK_BF16_TFLOPS,
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
        if world == 1 and not args.no_bs1:   # (behind the secondary rows: its graphs make streams of their own -- pipeline.pooled_stream)
            out['bs1_eval'] = bs1_eval(model, x, dev)
        print(json.dumps(out))
        if failed:
            sys.stdout.flush()
            raise SystemExit('bench.py: ' + failed)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
