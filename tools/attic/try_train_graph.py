"""Experiment: the stage-1 training step captured in ONE HIP graph (torch.cuda.CUDAGraph) and replayed -- does capture go through
the ctypes launches of the library, and what do 825 launches per step cost the host?   python tools/attic/try_train_graph.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import sc2bench_amd as S  # noqa: E402
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import training as T  # noqa: E402
from sc2bench_amd.resnet import resnet50  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
student = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False).to(dev)
teacher = resnet50().to(dev)
import copy
st_cfg = copy.deepcopy(bench.STAGE1)
st_cfg['optimizer']['kwargs']['capturable'] = True
stage = T.DistillationStage(teacher, student, st_cfg, dev, head_dtype=torch.bfloat16)
x = bench.synthetic_batch(256, dev, seed=0)


def step():
    loss = stage.forward_process(x, None)
    stage.post_forward_process(loss, bottleneck_updated=False)
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        l0 = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    l0 = step()
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t0) / 5 * 1e3, 'loss', float(l0))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        lg = step()
except Exception as e:   # noqa: BLE001
    import traceback
    traceback.print_exc()
    sys.exit(0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print('graph ms/step', (time.perf_counter() - t0) / 10 * 1e3, 'loss', float(lg))
