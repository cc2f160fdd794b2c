"""In-kernel clock of the bottleneck's launches, and what fraction of the matrix pipe's cycles each uses AT THAT CLOCK
(MI355X_MICROARCH.md "DVFS give-back" item 6).

For every launch of tools/k_times.py: >= 2 s of back-to-back launches on random data, then -- still back to back -- the library's
clock probe (csrc/diag.hip: sc2_clock_probe, one scalar wave per workgroup stamping s_memtime / s_memrealtime every 20 us) runs
beside it on a second stream for ~2.5 ms.  Per XCD: clock = delta s_memtime / delta s_memrealtime x 100 MHz over the probe's
life.  With the launch's duration t (HIP events over the same back-to-back run) and its algorithmic FLOPs:

    pipe_busy = FLOPs / (256 CUs x 4 SIMDs x 1 024 FLOP per cycle x clock x t)        (16 384 FLOP per 16-cycle v_mfma_f32_16x16x32_bf16)

= the share of matrix-pipe cycles that carry this launch's arithmetic at the clock the chip really holds; `of 2.5 PF` is the same
launch against the 2.4 GHz datasheet peak.  A launch with pipe_busy near 1 at a clock far below 2.4 GHz is at the chip's sustained
matrix rate: nothing in the kernel can raise it.  A launch with a low pipe_busy stalls.

    python tools/clock_probe.py [--bs 256] [--seconds 2.0] [--only dec] [--f32] [--head]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import k_times  # noqa: E402
from sc2bench_amd import hip  # noqa: E402


def probe_clock(fn, seconds, n_wg=16, n_samples=128, period_us=20.0):
    """-> (ms per launch, {xcc: MHz}, idle-chip sanity)"""
    dev = torch.device('cuda:0')
    side = torch.cuda.Stream(device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:      # DVFS steady state
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        fn()
    with torch.cuda.stream(side):
        samples = hip.clock_probe(n_wg, n_samples, period_us, stream=side)
    e0.record()
    iters = 0
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 1e-6 * period_us * n_samples * 1.5 and iters < 4000:
        fn()
        iters += 1
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / max(1, iters)
    s = samples.cpu()
    clocks = {}
    for w in range(n_wg):
        xcc = int(s[w, 0, 2])
        dt = (s[w, -1, 0] - s[w, 4, 0]).item()       # (the first samples may predate the kernels under study)
        dr = (s[w, -1, 1] - s[w, 4, 1]).item()
        if dr > 0:
            clocks.setdefault(xcc, []).append(100.0 * dt / dr)
    return ms, {k: sum(v) / len(v) for k, v in sorted(clocks.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=256)
    ap.add_argument('--seconds', type=float, default=2.0)
    ap.add_argument('--only', default='')
    ap.add_argument('--f32', action='store_true')
    ap.add_argument('--head', action='store_true')
    ap.add_argument('--head-layers', action='store_true', help='one representative launch per layer kind of the task head instead of the bottleneck rows')
    ap.add_argument('--zero-data', action='store_true', help='all-zero activations (torch.rand / randn patched to zeros while the inputs are built): the clock the chip holds when the operands do not toggle -- the power test of the sustained-rate reading')
    args = ap.parse_args()
    if args.zero_data:
        _rand, _randn = torch.rand, torch.randn
        torch.rand = lambda *a, **k: torch.zeros(*a, **{q: v for q, v in k.items() if q != 'generator'})
        torch.randn = lambda *a, **k: torch.zeros(*a, **{q: v for q, v in k.items() if q != 'generator'})
    rows = k_times.head_layer_rows(args.bs) if args.head_layers else k_times.build_rows(args.bs, args.only, args.head, args.f32)
    if args.head_layers and args.only:
        rows = [r for r in rows if args.only in r[0]]
    if args.zero_data:
        torch.rand, torch.randn = _rand, _randn
        print('ALL-ZERO activations')
    # idle chip: the probe alone
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s = hip.clock_probe(16, 64, 20.0, stream=side)
    torch.cuda.synchronize()
    s = s.cpu()
    idle = [100.0 * (s[w, -1, 0] - s[w, 4, 0]).item() / max(1, (s[w, -1, 1] - s[w, 4, 1]).item()) for w in range(16)]
    print('idle chip (probe alone): {:.0f} - {:.0f} MHz over {} XCDs seen'.format(min(idle), max(idle), len(set(int(s[w, 0, 2]) for w in range(16)))))
    print('{:<34}{:>9}{:>11}{:>10}{:>12}{:>11}   clock per XCD (MHz)'.format('launch', 'ms', 'TFLOP/s', 'of 2.5PF', 'clock MHz', 'pipe_busy'))
    for name, fn, flops in rows:
        if flops <= 0:
            continue
        ms, clocks = probe_clock(fn, args.seconds)
        if not clocks:
            print('{:<34} no probe samples'.format(name))
            continue
        clk = sum(clocks.values()) / len(clocks)
        tf = flops / ms / 1e9
        busy = flops / (256 * 4 * 1024 * clk * 1e6 * ms * 1e-3)
        f32 = 'f32' in name
        if f32:     # v_mfma_f32_16x16x4_f32: 2 048 FLOP per 32 cycles per SIMD = 64 FLOP per cycle (1 / 16 of the bf16 rate)
            busy = flops / (256 * 4 * 64 * clk * 1e6 * ms * 1e-3)
        print('{:<34}{:>9.4f}{:>11.1f}{:>10.3f}{:>12.0f}{:>11.3f}   {}'.format(
            name, ms, tf, tf / (157.3 if f32 else 2500.0), clk, busy, ' '.join('{}:{:.0f}'.format(k, v) for k, v in clocks.items())))


if __name__ == '__main__':
    main()
