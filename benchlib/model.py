"""The headline workload of bench.py: the Entropic-Student ResNet-50 (FP bottleneck, 24 channels) at 224 x 224 -- the roofline
constants, the per-launch algorithmic work table, the deterministic operating point and the synthetic batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def sha256_of(streams):
    import hashlib
    h = hashlib.sha256()
    for s in streams:
        h.update(len(s).to_bytes(4, 'little'))
        h.update(s)
    return h.hexdigest()
