"""Development aid: torch.profiler view of the stage-1 training step (which aten ops the step still launches around the HIP
kernels, and the host time per step):  python tools/attic/train_profile.py [--steps 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402
import sc2bench_amd as S  # noqa: E402
from sc2bench_amd import training as T  # noqa: E402
from sc2bench_amd.resnet import resnet50  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--bs', type=int, default=256)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
student = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False).to(dev)
teacher = resnet50().to(dev)
stage = T.DistillationStage(teacher, student, B.STAGE1, dev, head_dtype=torch.bfloat16)
x = B.synthetic_batch(args.bs, dev, seed=0)


def step():
    loss = stage.forward_process(x, None)
    stage.post_forward_process(loss, bottleneck_updated=False)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print('5 steps: issued in {:.1f} ms, done in {:.1f} ms per step'.format(1e3 * t_issue / 5, 1e3 * (time.perf_counter() - t0) / 5))
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=70))
