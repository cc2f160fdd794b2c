"""What the pipeline's concurrency costs the MFMA stages: wall time of the back stage (dequantise + decoder + head, bs 256), ten
calls back to back on one stream, (a) alone, (b) beside a range-coder launch (8 x 256 streams) on a second stream, (c) beside front
stages on a second stream, (d) beside both, (e) beside the library's clock probe (one scalar wave per workgroup, nothing else).
HIP events on the back stream; the side work is sized to cover the measured calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from sc2bench_amd import hip
dev = torch.device('cuda:0')
m = B.build_model(dev)
x = B.synthetic_batch(256, dev)
s_back, s_coder, s_front = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    sym, hw = m.stage_front(x)
    sym8 = torch.cat([sym] * 8)
    dec, _, _ = m.stage_coder(sym, hw, dequantized=True)
    for _ in range(2):
        m.stage_back(dec, hw); m.stage_coder(sym8, hw, dequantized=True)
    torch.cuda.synchronize()

    def run(coder=False, front=False, probe=False, n=10):
        torch.cuda.synchronize()
        if probe:
            with torch.cuda.stream(s_coder):
                hip.clock_probe(16, 2500, 20.0, stream=s_coder)      # 50 ms
        if coder:
            with torch.cuda.stream(s_coder):
                for _ in range(3):
                    m.stage_coder(sym8, hw, dequantized=True)
        if front:
            with torch.cuda.stream(s_front):
                for _ in range(2 * n):
                    m.stage_front(x)
        with torch.cuda.stream(s_back):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_back)
            for _ in range(n):
                m.stage_back(dec, hw)
            e1.record(s_back)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for rnd in range(3):
        print('round {}: alone {:.3f} | + coder {:.3f} | + fronts {:.3f} | + both {:.3f} | + probe waves only {:.3f}   ms per back stage'.format(
            rnd, run(), run(coder=True), run(front=True), run(coder=True, front=True), run(probe=True)))
    # the same for the front stage
    def runf(coder=False, n=20):
        torch.cuda.synchronize()
        if coder:
            with torch.cuda.stream(s_coder):
                for _ in range(2):
                    m.stage_coder(sym8, hw, dequantized=True)
        with torch.cuda.stream(s_front):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_front)
            for _ in range(n):
                m.stage_front(x)
            e1.record(s_front)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    print('front stage: alone {:.3f} | + coder {:.3f} ms'.format(runf(), runf(True)))

# ---- per launch: duration alone vs beside coder launches (which launches pay for the coder's waves?)
print()
print('per launch, ms alone -> beside three 2 048-stream coder launches (HIP events around each tagged launch of the back stage, 10 calls)')
with torch.no_grad():
    def tagged(coder):
        torch.cuda.synchronize()
        if coder:
            with torch.cuda.stream(s_coder):
                for _ in range(3):
                    m.stage_coder(sym8, hw, dequantized=True)
        with torch.cuda.stream(s_back):
            with hip.KernelTimer() as t:
                for _ in range(10):
                    m.stage_back(dec, hw)
                torch.cuda.synchronize()
        return {k: v[1] for k, v in t.summary().items()}
    a, b = tagged(False), tagged(True)
    a2, b2 = tagged(False), tagged(True)
    rows = sorted(a, key=lambda k: -(b[k] + b2[k] - a[k] - a2[k]))
    tot_a = tot_b = 0.0
    for k in rows:
        ka, kb = 0.5 * (a[k] + a2[k]), 0.5 * (b[k] + b2[k])
        tot_a += ka; tot_b += kb
        print('{:<28} {:7.3f} -> {:7.3f}  ({:+.3f}, x{:.2f})'.format(k, ka, kb, kb - ka, kb / ka))
    print('{:<28} {:7.3f} -> {:7.3f}'.format('sum', tot_a, tot_b))
