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


# ---------------------------------------------------------------------------------------------------------------
# DP == single process on the concatenated batch, through DistillationStage
STAGE = {
    'teacher': {'sequential': ['layer1', 'layer2'], 'forward_hook': {'input': [], 'output': ['layer1', 'layer2']}},
    'student': {'sequential': ['bottleneck_layer', 'layer2'], 'frozen_modules': ['layer2'],
                'forward_hook': {'input': [], 'output': ['bottleneck_layer', 'layer2', 'bottleneck_layer.entropy_bottleneck']}},
    'optimizer': {'key': 'SGD', 'kwargs': {'lr': 0.0}},
    'criterion': {'key': 'WeightedSumLoss', 'kwargs': {'sub_terms': {
        'layer1': {'criterion': {'key': 'MSELoss', 'kwargs': {'reduction': 'sum'}},
                   'criterion_wrapper': {'key': 'SimpleLossWrapper', 'kwargs': {
                       'input': {'is_from_teacher': False, 'module_path': 'bottleneck_layer', 'io': 'output'},
                       'target': {'is_from_teacher': True, 'module_path': 'layer1', 'io': 'output'}}}, 'weight': 1.0},
        'layer2': {'criterion': {'key': 'MSELoss', 'kwargs': {'reduction': 'sum'}},
                   'criterion_wrapper': {'key': 'SimpleLossWrapper', 'kwargs': {
                       'input': {'is_from_teacher': False, 'module_path': 'layer2', 'io': 'output'},
                       'target': {'is_from_teacher': True, 'module_path': 'layer2', 'io': 'output'}}}, 'weight': 1.0},
        'bpp': {'criterion': {'key': 'BppLoss', 'kwargs': {'entropy_module_path': 'bottleneck_layer.entropy_bottleneck',
                                                           'reduction': 'sum'}}, 'weight': 0.08}}}},
}


def _build_pair():
    """CPU-safe student (the oracle's FP bottleneck with a per-image deterministic noise draw + a frozen conv tail) and
    teacher, identical on every rank."""
    import torch.nn as nn
    from oracle import cpu_ref as R

    class DetEB(R.EntropyBottleneck):
        def forward(self, x, training=None, noise=None):
            g = torch.Generator().manual_seed(99)
            base = torch.rand(x.shape[1:], generator=g) - 0.5          # the same noise field for every image
            return super().forward(x, training, noise=base.unsqueeze(0).expand_as(x))

    class Student(nn.Module):
        def __init__(self):
            super().__init__()
            self.bottleneck_layer = R.FPBasedResNetBottleneck(num_bottleneck_channels=8, num_target_channels=16)
            eb = DetEB(8)
            eb.load_state_dict(self.bottleneck_layer.entropy_bottleneck.state_dict())
            self.bottleneck_layer.entropy_bottleneck = eb
            self.layer2 = nn.Conv2d(16, 8, 3, padding=1)

        def get_aux_module(self):
            return self.bottleneck_layer

    class Teacher(nn.Module):
        def __init__(self):
            super().__init__()
            self.layer1 = nn.Conv2d(3, 16, 5, stride=4, padding=2)
            self.layer2 = nn.Conv2d(16, 8, 3, padding=1)

    torch.manual_seed(5)
    return Teacher(), Student()


def _grads_after_step(batch, distributed):
    from sc2bench_amd import training as T
    teacher, student = _build_pair()
    stage = T.DistillationStage(teacher, student, STAGE, torch.device('cpu'), bucket_mb=0.02)
    if distributed:
        assert stage.reducer.world == 2 and len(stage.reducer.flats) >= 3
    loss = stage.forward_process(batch)
    # post_forward_process without the optimizer step / zero_grad, so that the reduced gradients can be read
    stage.aux_module.aux_loss().backward()
    with stage.reducer.overlap():
        loss.backward()
    hooked = stage.reducer.launched_by_hook
    stage.reducer.all_reduce()
    return [p.grad.clone() for p in stage.reducer.params], float(loss.detach()), hooked


def _dp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    from sc2bench_amd import dataparallel as dp
    dp.init_distributed(backend='gloo')
    torch.set_num_threads(2)
    x = torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    s, e = dp.shard_range(4, rank, world)
    grads, loss, hooked = _grads_after_step(x[s:e], True)
    out[rank] = ([g.numpy() for g in grads], loss, hooked)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradients_equal_single_process_on_concatenated_batch():
    """Losses are sums over the local batch (MSE sum, bits sum, as in the ES recipe), so the rank-averaged DP gradient
    times the world size is the gradient of the single-process loss on the concatenated batch; the aux-loss gradient
    (batch-independent, identical on every rank) survives the average unchanged."""
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_dp_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
            assert p.exitcode == 0
        res = dict(out)
    torch.set_num_threads(2)
    x = torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    single, loss, _ = _grads_after_step(x, False)
    # the same step with the aux-loss gradient alone (batch-independent part)
    from sc2bench_amd import training as T
    teacher, student = _build_pair()
    stage = T.DistillationStage(teacher, student, STAGE, torch.device('cpu'))
    stage.aux_module.aux_loss().backward()
    aux = [p.grad.clone() for p in stage.reducer.params]
    assert abs(res[0][1] + res[1][1] - loss) <= 1e-4 * abs(loss)
    assert res[0][2] >= 1, 'no bucket was launched from a gradient hook during backward'
    for g0, g1, gs, ga in zip(res[0][0], res[1][0], single, aux):
        g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
        assert torch.equal(g0, g1), 'ranks disagree after the all-reduce'
        want = (gs - ga) / world + ga
        tol = 1e-4 * float(want.abs().max()) + 1e-6
        assert float((g0 - want).abs().max()) <= tol


def test_bench_torchrun_dry_run_world2():
    """bench.py under torch.distributed.run with 2 ranks, as the driver launches it: rank/shard/seed plumbing, the
    barrier and the max-over-ranks reduction, on gloo; --dry-run exits before any GPU call."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--dry-run']
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line from rank 0: {}'.format(r.stdout)
    rec = json.loads(lines[0])
    assert rec['dry_run'] and rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['warmup'] == 1
    assert rec['config']['global_batch'] == 2 * rec['config']['batch_per_gpu']
    assert [{k: q[k] for k in ('rank', 'local_rank', 'seed')} for q in rec['ranks']] == \
        [{'rank': 0, 'local_rank': 0, 'seed': 0}, {'rank': 1, 'local_rank': 1, 'seed': 1}]
    assert [q['shard'] for q in rec['ranks']] == [[0, 256], [256, 512]] and rec['config']['ranks_reduced'] == 2
    assert rec['scaling'] == 'weak' and rec['value'] > 0


def test_bench_self_launch_dry_run_world2():
    """`bench.py --gpus 2 --dry-run` with NO launcher environment: bench.py itself starts the 2 ranks as child processes
    (torch.distributed.run, 127.0.0.1, a free port), relays rank 0's single JSON line and returns the children's code."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line from rank 0: {}'.format(r.stdout)
    rec = json.loads(lines[0])
    assert rec['dry_run'] and rec['n_gpus'] == 2 and rec['steps'] == 3
    assert [q['rank'] for q in rec['ranks']] == [0, 1]


def test_bench_self_launch_dry_run_world8():
    """The driver's 8-GPU launch shape on gloo (no device): `bench.py --gpus 8 --dry-run` starts eight ranks itself; the shards of
    the global batch are disjoint and cover it, an all-reduce on the backend counts eight ranks, and the reported time is the MAX
    over ranks (rank r sleeps (1 + r) ms per step: the line must carry rank 7's time, not rank 0's)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '5', '--warmup', '1', '--bs', '256', '--dry-run']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line from rank 0: {}'.format(r.stdout)
    rec = json.loads(lines[0])
    assert rec['dry_run'] and rec['n_gpus'] == 8 and rec['config']['global_batch'] == 2048 and rec['config']['ranks_reduced'] == 8
    ranks = rec['ranks']
    assert [q['rank'] for q in ranks] == list(range(8)) and sorted(q['seed'] for q in ranks) == list(range(8))
    shards = sorted(tuple(q['shard']) for q in ranks)
    assert shards[0][0] == 0 and shards[-1][1] == 2048
    assert all(a[1] == b[0] for a, b in zip(shards, shards[1:])), 'shards overlap or leave a gap: {}'.format(shards)
    slowest = max(q['own_elapsed_s'] for q in ranks)
    # (own_elapsed_s closes behind the barrier, so every rank's is about the slowest rank's time; own_work_s stops in front of it)
    assert slowest >= 5 * 0.008 and ranks[0]['own_work_s'] < ranks[7]['own_work_s'] <= slowest
    assert rec['ms_per_step'] * 5e-3 >= slowest - 1e-6, 'the line does not carry the MAX over ranks'
    assert abs(rec['value'] - 2048 * 5 / (rec['ms_per_step'] * 5e-3)) < 1e-6 * rec['value']
    # the per-rank fields a multi-GPU line keeps: every rank's own bpp and the digest of its first 8 byte
    # streams -- in the dry run from the library's host range coder on rank-seeded symbols, so eight DIFFERENT digests
    digests = [q['bitstream_sha256_first8'] for q in ranks]
    assert all(len(d) == 64 and int(d, 16) >= 0 for d in digests) and len(set(digests)) == 8
    assert all(q['rans_status'] == 0 and 0.5 < q['bpp'] < 20.0 for q in ranks)
    import numpy as np
    from oracle import rans as oracle_rans
    import bench as bench_mod
    # rank 3's streams again, here, through the ORACLE's coder: the digest the rank reported is that of the reference algorithm's bytes
    t = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'rans_kat.json')))['table']
    width = max(len(r) for r in t['cdfs'])
    cdf = np.zeros((len(t['cdfs']), width), np.int32)
    for i, r in enumerate(t['cdfs']):
        cdf[i, :len(r)] = r
    sym = np.random.RandomState(1003).randint(-3, 4, size=(8, 24 * 55 * 55)).astype(np.int32)
    idx = np.zeros(sym.shape[1], np.int32)
    ref = [oracle_rans.encode_with_indexes(sym[i], idx, cdf, np.array(t['cdf_sizes'], np.int32), np.array(t['offsets'], np.int32))
           for i in range(8)]
    assert bench_mod.sha256_of(ref) == ranks[3]['bitstream_sha256_first8']


def test_bench_refuses_gpus_world_mismatch():
    """--gpus must equal the world size the launcher made: a line that says n_gpus 1 for `--gpus 2` is not a result."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', OMP_NUM_THREADS='1')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--dry-run']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE 1' in r.stderr


def _eval_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    from sc2bench_amd import dataparallel as dp, evaluation as E
    dp.init_distributed(backend='gloo')
    torch.manual_seed(0)
    model = torch.nn.Linear(8, 10)        # every rank holds the same classifier
    g = torch.Generator().manual_seed(3)
    xs, ys = torch.randn(10, 8, generator=g), torch.randint(0, 10, (10,), generator=g)
    lo, hi = (0, 7) if rank == 0 else (7, 10)    # UNEQUAL shards: 7 and 3 samples
    loader = [(xs[i:i + 1], ys[i:i + 1]) for i in range(lo, hi)]
    res = E.evaluate(model, loader, torch.device('cpu'), log_freq=0)
    out[rank] = (res['acc1'], res['acc5'], res['samples'], res['samples_all_ranks'])
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_world2_sums_totals_and_counts():
    """evaluate() under a process group: the metric is total / count over ALL ranks' samples (the reference's
    MetricLogger.synchronize_between_processes), not the mean of the per-rank averages -- the two differ when ranks
    saw different numbers of samples; the reduction runs on the backend's device."""
    from sc2bench_amd import evaluation as E
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_eval_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
            assert p.exitcode == 0
        res = dict(out)
    torch.manual_seed(0)
    model = torch.nn.Linear(8, 10)
    g = torch.Generator().manual_seed(3)
    xs, ys = torch.randn(10, 8, generator=g), torch.randint(0, 10, (10,), generator=g)
    single = E.evaluate(model, [(xs[i:i + 1], ys[i:i + 1]) for i in range(10)], torch.device('cpu'), log_freq=0)
    for r in range(world):
        assert abs(res[r][0] - single['acc1']) < 1e-9 and abs(res[r][1] - single['acc5']) < 1e-9
        assert res[r][3] == 10
    assert (res[0][2], res[1][2]) == (7, 3)


def test_rank_binds_to_its_gpus_numa_node(tmp_path, monkeypatch):
    """dataparallel.bind_rank_to_gpu_numa on a fake sysfs tree (two CPU nodes + four GPUs, two per socket): rank r lands on the CPUs
    of the socket its GPU hangs off, HIP_VISIBLE_DEVICES re-maps the index, an unknown topology binds nothing; no HIP call."""
    import sc2bench_amd.dataparallel as dp
    root = tmp_path
    nodes = root / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes'
    cpus_all = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus_all) // 2)
    cpu_sets = [cpus_all[:half], cpus_all[half:] or cpus_all[:half]]
    for n in (0, 1):          # CPU nodes: no SIMDs
        (nodes / str(n)).mkdir(parents=True)
        (nodes / str(n) / 'properties').write_text('cpu_cores_count 4\nsimd_count 0\nlocation_id 0\ndomain 0\n')
        nd = root / 'devices' / 'system' / 'node' / 'node{}'.format(n)
        nd.mkdir(parents=True)
        nd.joinpath('cpulist').write_text(','.join(str(c) for c in cpu_sets[n]) + '\n')
    for g in range(4):        # GPU g at bus 0x10 * (g + 1), device 0, function 0; GPUs 0, 1 on socket 0, GPUs 2, 3 on socket 1
        bus = 0x10 * (g + 1)
        (nodes / str(2 + g)).mkdir()
        (nodes / str(2 + g) / 'properties').write_text('cpu_cores_count 0\nsimd_count 1024\nlocation_id {}\ndomain 0\n'.format(bus << 8))
        pd = root / 'bus' / 'pci' / 'devices' / '0000:{:02x}:00.0'.format(bus)
        pd.mkdir(parents=True)
        pd.joinpath('numa_node').write_text('{}\n'.format(g // 2))
    monkeypatch.delenv('HIP_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    assert [dp.gpu_numa_node(r, str(root)) for r in range(4)] == [0, 0, 1, 1] and dp.gpu_numa_node(4, str(root)) is None
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '3,0')
    assert dp.gpu_numa_node(0, str(root)) == 1 and dp.gpu_numa_node(1, str(root)) == 0
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    before = os.sched_getaffinity(0)
    try:
        got = dp.bind_rank_to_gpu_numa(2, str(root))
        assert got == {'numa_node': 1, 'cpus': len(set(cpu_sets[1]))} and os.sched_getaffinity(0) == set(cpu_sets[1])
    finally:
        os.sched_setaffinity(0, before)
    assert dp.bind_rank_to_gpu_numa(0, str(tmp_path / 'nowhere'), str(tmp_path / 'nodev')) is None and os.sched_getaffinity(0) == before
    # second source: the KFD properties are unreadable (as for a non-root user on the pool's boxes) -- the render nodes this process
    # can open, resolved to PCI addresses: a container that was given GPUs 2 and 3 only
    import shutil
    shutil.rmtree(str(nodes))
    dri = tmp_path / 'dev' / 'dri'
    dri.mkdir(parents=True)
    for g in (2, 3):
        bus = 0x10 * (g + 1)
        (dri / 'renderD{}'.format(128 + g)).write_text('')
        link = root / 'class' / 'drm' / 'renderD{}'.format(128 + g)
        link.mkdir(parents=True)
        (root / 'bus' / 'pci' / 'devices' / '0000:{:02x}:00.0'.format(bus) / 'vendor').write_text('0x1002\n')
        os.symlink(str(root / 'bus' / 'pci' / 'devices' / '0000:{:02x}:00.0'.format(bus)), str(link / 'device'))
    (dri / 'card0').write_text('')
    assert [dp.gpu_numa_node(r, str(root), str(tmp_path / 'dev')) for r in range(3)] == [1, 1, None]
