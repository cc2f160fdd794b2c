"""Range-coder launch time vs escape rate: 2 048 streams x 72 600 symbols (the bench's coder launch), 24 rows of 10-19 symbols,
in-table symbols uniform, a fraction `p` of the symbols outside the table (bypass-coded).  HIP events, one stream."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S  # noqa: E402
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from oracle import rans as oracle_rans  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    rows, sizes, offs = [], [], []
    for r in range(24):
        n = 10 + r % 10
        p = rng.rand(n).astype(np.float32) ** 2 + 1e-3
        p /= p.sum()
        cdf = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-(n // 2))
    width = max(len(r) for r in rows)
    cdfs = torch.tensor([r + [0] * (width - len(r)) for r in rows], dtype=torch.int32, device=dev)
    d_sizes = torch.tensor(sizes, dtype=torch.int32, device=dev)
    d_offs = torch.tensor(offs, dtype=torch.int32, device=dev)
    n_streams, hw = int(os.environ.get('STREAMS', 2048)), 3025
    n_sym = 24 * hw
    g = torch.Generator(device=dev).manual_seed(0)
    row_of = (torch.arange(n_sym, device=dev) // hw)
    nmax = (d_sizes - 2)[row_of]
    off = d_offs[row_of]
    for p_esc in (0.0, 1e-5, 1e-4, 1e-3, 1e-2, 0.1):
        u = torch.rand((n_streams, n_sym), device=dev, generator=g)
        sym = (u * nmax).floor().int() + off                          # in table
        esc = torch.rand((n_streams, n_sym), device=dev, generator=g) < p_esc
        sym = torch.where(esc, nmax.int() + off + 3 + (u * 40).int(), sym).contiguous()
        res = []
        for it in range(3):
            t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t0.record()
            buf, o, nb, st = S.hip.rans_encode_batch(sym, cdfs, d_sizes, d_offs, index_div=hw)
            t1.record()
            dec, dst = S.hip.rans_decode_batch(buf, o, nb, n_sym, cdfs, d_sizes, d_offs, index_div=hw)
            t2.record()
            torch.cuda.synchronize()
            res.append((t0.elapsed_time(t1), t1.elapsed_time(t2)))
        assert int(st.max()) == 0 and torch.equal(dec, sym)
        e, d = min(r[0] for r in res), min(r[1] for r in res)
        print('escape rate {:<8g} encode {:7.2f} ms  decode {:7.2f} ms  ({:.0f} / {:.0f} ns per symbol per stream)  bytes/stream {:.0f}'
              .format(p_esc, e, d, e * 1e6 / n_sym, d * 1e6 / n_sym, nb.float().mean().item()))


if __name__ == '__main__':
    main()
