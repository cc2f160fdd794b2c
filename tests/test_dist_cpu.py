"""N > 1 path on CPU: world_size-2 gloo process group, flat-bucket gradient averaging, parameter broadcast and
image sharding -- the host logic bench.py / training use under RCCL on the GPU box."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    import sc2bench_amd as S
    from sc2bench_amd import dataparallel as dp
    distributed, r, w, device = dp.init_distributed(backend='gloo')
    assert distributed and r == rank and w == world and device.type == 'cpu'
    torch.manual_seed(100 + rank)                 # different init per rank ...
    m = S.FPBasedResNetBottleneck()
    dp.broadcast_parameters(m, src=0)             # ... equalised by the broadcast
    checksum = sum(float(p.double().sum()) for p in m.parameters())
    for p in m.decoder.parameters():              # stage-1-like: a frozen subset is left out of the buckets
        p.requires_grad_(False)
    red = dp.FlatGradAllReducer(m.parameters(), bucket_mb=0.25)
    assert len(red.flats) > 1 and red.nbytes() == 4 * sum(p.numel() for p in m.parameters() if p.requires_grad)
    gen = torch.Generator().manual_seed(7)
    expect = []
    for p in red.params:                          # synthetic per-rank gradients with a known mean
        base = torch.randn(p.shape, generator=gen)
        p.grad.copy_(base * (rank + 1))
        expect.append(base * (sum(range(1, world + 1)) / world))
    red.all_reduce()
    err = max(float((p.grad - e).abs().max()) for p, e in zip(red.params, expect))
    s, e = dp.shard_range(11, rank, world)
    mean_metric = dp.all_reduce_mean_scalars([float(rank), 1.0], device)
    out[rank] = (checksum, err, (s, e), mean_metric, all(p.grad.data_ptr() != 0 for p in red.params))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradient_average_and_sharding():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
        res = dict(out)
    assert abs(res[0][0] - res[1][0]) < 1e-9, 'broadcast did not equalise parameters'
    assert res[0][1] < 1e-6 and res[1][1] < 1e-6, 'flat-bucket all-reduce is not the mean'
    assert res[0][2] == (0, 6) and res[1][2] == (6, 11)
    assert res[0][3] == [0.5, 1.0] and res[1][3] == [0.5, 1.0]


def test_shard_range_covers_everything():
    from sc2bench_amd import dataparallel as dp
    for n in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
