"""The launch contract of bench.py without a device (`--dry-run`: ranks, shards, barrier, MAX over ranks, per-rank digests on
gloo) and the self-launcher of `bench.py --gpus N`."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

from .model import ROOT, sha256_of


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
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print('bench.py: the {}-rank launch failed with return code {}'.format(n, rc), file=sys.stderr)
    sys.exit(rc)
