"""RCCL on the hardware a one-GPU box has: the launch contract of bench.py with a REAL `nccl` (= RCCL) process group of one
rank.  `python -m torch.distributed.run --nproc-per-node 1 bench.py ...` runs as a CHILD process, so

  * `init_process_group('nccl', device_id=...)`, the barriers around the timed region and the MAX all-reduce of the elapsed
    time execute on RCCL (inference line), and
  * with SC2_DP_WORLD1_COLLECTIVES=1 the training step issues every collective of the data-parallel path on HIP tensors:
    `broadcast_parameters`, `FlatGradAllReducer`'s bucket all-reduces launched from post-accumulate-grad hooks inside
    backward, `all_reduce_sum_scalars` on the backend's device (script/task/image_classification.py:110-111 wraps the student
    in DistributedDataParallel for the same purpose).

The file name sorts first on purpose: the child is started BEFORE anything in the pytest process has initialised the GPU
(a process that has must not fork + exec on this pool); if something already has, the test skips and says so.  No scaling
curve is measured here -- that needs more than one GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _torchrun_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    env.update(env_extra or {})
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, 'rc {}\nstdout: {}\nstderr: {}'.format(r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_rccl_world1_inference_and_training_as_child_processes():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.skip('no HIP device')
    if torch.cuda.is_initialized():
        pytest.skip('this process has already initialised the GPU: the RCCL child processes must be started first '
                    '(run the whole suite, or this file alone)')
    # ---- inference line: process group, barriers, MAX all-reduce of the timing
    line = _torchrun_bench(['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-bs1'])
    assert line['n_gpus'] == 1 and line['steps'] == 2 and line['unit'] == 'images/s' and line['value'] > 0
    assert line['config']['process_group'].startswith('nccl'), line['config']['process_group']
    assert line['config']['ranks_reduced'] == 1, 'the device all-reduce of 1 over the RCCL group must count this rank'
    assert line['scaling'] == 'weak' and line['config']['sharding'] == 'images, no collective'
    # every rank's own bpp and stream digest, gathered as device tensors on the backend: here one rank, whose
    # row must be the line's own figures
    assert line['config']['ranks_gathered'] == 1 and len(line['per_rank']) == 1
    row = line['per_rank'][0]
    assert row['rank'] == 0 and row['rans_status'] == 0 and abs(row['bpp'] - line['bpp']) < 1e-9
    assert row['bitstream_sha256_first8'] == line['bitstream_sha256_first8'] and len(row['bitstream_sha256_first8']) == 64
    # ---- training step: broadcast, hook-launched bucket all-reduces, scalar reductions, all on RCCL with HIP tensors
    tr = _torchrun_bench(['--mode', 'train', '--steps', '2', '--warmup', '1', '--bs', '32'],
                         env_extra={'SC2_DP_WORLD1_COLLECTIVES': '1'})
    cfg = tr['config']
    assert cfg['process_group'].startswith('nccl') and cfg['collectives_issued'] is True
    assert cfg['gradient_buckets'] >= 1 and cfg['buckets_launched_from_backward_hooks_last_step'] >= 1
    assert tr['images_all_ranks'] == 64.0                      # all_reduce_sum_scalars on the device: 2 steps x 32 images x 1 rank
    assert tr['value'] > 0 and tr['final_loss'] == tr['final_loss']   # finite
