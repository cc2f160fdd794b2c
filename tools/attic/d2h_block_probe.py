"""Does a second non_blocking device-to-host copy block its caller while the first is in flight (one stream / two streams / with a
thread waiting on the first copy's event)?"""
import threading, time, torch
dev = torch.device('cuda:0')
n = 256 * 72600
src = [torch.randint(0, 10, (n,), dtype=torch.int32, device=dev) for _ in range(3)]
dst = [torch.empty(n, dtype=torch.int32, pin_memory=True) for _ in range(3)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def go(streams, waiter):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out = []
    evs = []
    for k in range(3):
        with torch.cuda.stream(streams[k]):
            a = time.perf_counter()
            dst[k].copy_(src[k], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(streams[k]); evs.append(ev)
            out.append(round(1e3 * (time.perf_counter() - a), 3))
        if k == 0 and waiter:
            th = threading.Thread(target=lambda: evs[0].synchronize()); th.start()
    torch.cuda.synchronize()
    return out, round(1e3 * (time.perf_counter() - t0), 2)
for name, streams, waiter in (('one stream', [s1, s1, s1], False), ('two streams', [s1, s2, s1], False), ('one stream + a thread waiting on copy 0', [s1, s1, s1], True),
                              ('one stream again', [s1, s1, s1], False)):
    for _ in range(2):
        print(name, 'host ms per copy call', *go(streams, waiter))

# ---- the same copies while a worker thread runs the host range coder (64 threads spawned per call)
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sc2bench_amd import hip
g = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'fp_golden.pt'), weights_only=False)
cdf, ln, off = g['quantized_cdf'].numpy(), g['cdf_length'].numpy().reshape(-1), g['offset'].numpy().reshape(-1)
tb = hip.HostRansTables(cdf, ln, off)
rng = np.random.RandomState(0)
hw = 55 * 55
one = np.concatenate([np.clip(np.round(rng.randn(hw) * 1.5).astype(np.int32), off[c], off[c] + ln[c] - 3) for c in range(24)]).astype(np.int32)
sym = np.ascontiguousarray(np.tile(one, (256, 1)))
dec = np.empty_like(sym)
scratch = {}
hip.rans_code_host(tb, sym, hw, dec, scratch=scratch)
for thr in (64, 16):
    for _ in range(3):
        th = threading.Thread(target=lambda: hip.rans_code_host(tb, sym, hw, dec, scratch=scratch, threads=thr))
        torch.cuda.synchronize()
        th.start()
        time.sleep(0.0005)
        a = time.perf_counter(); out = []
        for k in range(3):
            with torch.cuda.stream(s1):
                b = time.perf_counter()
                dst[k].copy_(src[k], non_blocking=True)
                out.append(round(1e3 * (time.perf_counter() - b), 3))
        th.join()
        torch.cuda.synchronize()
        print('copies beside the host coder on {} threads: host ms per copy call'.format(thr), out)
