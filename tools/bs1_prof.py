"""Where a bs-1 evaluation forward (encode -> host bytes -> decode -> head) spends its time, and what HIP-graph replay of its two
device halves would cost: sections timed with a synchronize between them (so they add up to more than the un-synchronised forward)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import sc2bench_amd as S
from sc2bench_amd import hip

dev = torch.device('cuda:0')
model = bench.build_model(dev)
x = bench.synthetic_batch(8, dev, seed=0)
bl, eb = model.bottleneck_layer, model.bottleneck_layer.entropy_bottleneck
N = 64


def timed(fn, n=N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad():
    for i in range(3):
        model(x[i:i + 1])
    print('forward()                 {:.3f} ms / image'.format(timed(lambda i: model(x[i % 8:i % 8 + 1]))))
    one = x[0:1]
    sym = bl.analysis(one, symbols_for=eb)
    shape = tuple(sym.shape[-2:])
    hw = shape[0] * shape[1]
    print('analysis (3 launches)     {:.3f}'.format(timed(lambda i: bl.analysis(one, symbols_for=eb))))
    sym2 = sym.reshape(1, -1)
    print('symbols D2H               {:.3f}'.format(timed(lambda i: sym2.cpu().numpy())))
    sym_h = sym2.cpu().numpy()
    tables = eb._host_tables()
    t0 = time.perf_counter()
    for _ in range(N):
        strings, st = hip.rans_encode_host(tables, sym_h, index_div=hw)
    print('host encode               {:.3f}'.format(1e3 * (time.perf_counter() - t0) / N))
    t0 = time.perf_counter()
    for _ in range(N):
        dec_h, st = hip.rans_decode_host(tables, strings, sym_h.shape[1], index_div=hw)
    print('host decode               {:.3f}'.format(1e3 * (time.perf_counter() - t0) / N))
    print('H2D + dequantise          {:.3f}'.format(timed(lambda i: hip.eb_dequantize(torch.from_numpy(dec_h).to(dev).view(1, -1, *shape), eb._median_vector(), want_f32=False, want_nhwc=True))))
    y_hat = hip.eb_dequantize(torch.from_numpy(dec_h).to(dev).view(1, -1, *shape), eb._median_vector(), want_f32=False, want_nhwc=True)[1]
    print('synthesis                 {:.3f}'.format(timed(lambda i: bl.synthesis_nhwc(y_hat))))
    feats = bl.synthesis_nhwc(y_hat)
    print('head                      {:.3f}'.format(timed(lambda i: model.head(feats))))
    print('decode_head (tail fused)  {:.3f}'.format(timed(lambda i: model.decode_head(y_hat))))

    # ---- HIP graphs of the two device halves
    x_static = one.clone()
    sym_static_in = torch.from_numpy(dec_h).to(dev).view(1, -1, *shape).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            bl.analysis(x_static, symbols_for=eb)
            model.decode_head(hip.eb_dequantize(sym_static_in, eb._median_vector(), want_f32=False, want_nhwc=True)[1])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gA = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA):
        sym_out = bl.analysis(x_static, symbols_for=eb)
    gB = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gB):
        logits_out = model.decode_head(hip.eb_dequantize(sym_static_in, eb._median_vector(), want_f32=False, want_nhwc=True)[1])
    torch.cuda.synchronize()
    print('graph A replay            {:.3f}'.format(timed(lambda i: gA.replay())))
    print('graph B replay            {:.3f}'.format(timed(lambda i: gB.replay())))
    ref_logits = model(one).float()
    gA.replay(); gB.replay(); torch.cuda.synchronize()
    print('graph outputs equal eager:', torch.equal(sym_out.reshape(-1).cpu(), sym.reshape(-1).cpu()), torch.equal(logits_out.float(), ref_logits))

    def graphed(i):
        x_static.copy_(x[i % 8:i % 8 + 1])
        gA.replay()
        sh = sym_out.reshape(1, -1).cpu().numpy()
        strings, st = hip.rans_encode_host(tables, sh, index_div=hw)
        d, st = hip.rans_decode_host(tables, strings, sh.shape[1], index_div=hw)
        sym_static_in.copy_(torch.from_numpy(d).view(sym_static_in.shape), non_blocking=False)
        gB.replay()
        return logits_out
    print('graphed forward           {:.3f} ms / image'.format(timed(graphed)))
