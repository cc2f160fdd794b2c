"""Development aid: the C x C GEMMs of GDN1 forward / backward at the training shapes (bs 256), with and without their epilogues --
how much of a launch is the tile kernel's main loop and how much the parked-tile epilogue:  python tools/gdn_gemm_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc2bench_amd import hip  # noqa: E402


def timeit(fn, iters=8, warmup=3):
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


dev = torch.device('cuda:0')
torch.manual_seed(0)
for C, hw, inverse in ((512, 56, True), (256, 55, True), (96, 112, False)):
    N = 256
    x = torch.randn(N, hw, hw, C, device=dev).to(torch.bfloat16)
    g = torch.randn(N, hw, hw, C, device=dev).to(torch.bfloat16)
    gamma = (torch.rand(C, C, device=dev) * 0.1 / C ** 0.5 + 0.1 * torch.eye(C, device=dev)).float()
    beta = torch.ones(C, device=dev)
    w = hip.pack_conv_weight(gamma.reshape(C, C, 1, 1))
    wt = hip.pack_conv_weight(gamma.t().reshape(C, C, 1, 1))
    gb = x.numel() * 2 / 1e9
    tf = 2.0 * x.numel() * C / 1e12
    rows = [
        ('plain GEMM (no epilogue)', lambda: hip.conv2d_fwd(x, w, C, 1, 1, 1, 0), 2),
        ('|x| GEMM + bias', lambda: hip.conv2d_fwd(x, w, C, 1, 1, 1, 0, a_op=hip.AOP_ABS, epilogue=hip.EPI_BIAS, ep_beta=beta), 2),
        ('forward GDN1 (ep_x)', lambda: hip.conv2d_fwd(x, w, C, 1, 1, 1, 0, a_op=hip.AOP_ABS, epilogue=hip.EPI_IGDN if inverse else hip.EPI_GDN,
                                                       ep_x=x, ep_beta=beta), 3),
        ('bwd PRE', lambda: hip.gdn1_bwd_gemm(x, w, hip.EPI_IGDN1_BWD_PRE if inverse else hip.EPI_GDN1_BWD_PRE, g, x, beta), 5),
        ('bwd POST', lambda: hip.gdn1_bwd_gemm(g, wt, hip.EPI_GDN1_BWD_POST, g, x), 4),
        ('colsum', lambda: hip.colsum_bf16(g, C), 1),
        ('wgrad |x|', lambda: hip.conv2d_wgrad(x, g, 1, 1, 1, 0, x_abs=True), 2),
    ]
    if hip.gdn1_rows_supported(x, C):
        gf, gtf = hip.pack_weight_fragments(gamma), hip.pack_weight_fragments(gamma.t().contiguous())
        rows += [('rows forward', lambda: hip.gdn1_rows_fwd(x, gf, beta, inverse), 2),
                 ('rows backward (both GEMMs)', lambda: hip.gdn1_rows_bwd(x, g, gf, gtf, beta, inverse), 4)]
    print('C = {} at {} x {} x {} ({:.3f} GB per tensor, {:.3f} TFLOP per GEMM)'.format(C, N, hw, hw, gb, tf))
    for name, fn, tensors in rows:
        ms = timeit(fn)
        print('  {:<28}{:8.3f} ms   {:7.1f} TFLOP/s   {:5.2f} TB/s over {} tensors'.format(name, ms, tf / ms * 1e3, tensors * gb / ms, tensors))
