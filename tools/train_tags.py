"""Development aid: where the stage-1 training step's GPU time goes, by launch tag (HIP-event pairs around every launch made through
`sc2bench_amd.hip`) and by phase (teacher forward / student forward / criterion / backward + step, event pairs on the current
stream):  python tools/train_tags.py [--steps 3] [--policy gdn_bwd_fused=0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import sc2bench_amd as S  # noqa: E402
from sc2bench_amd import hip, training as T  # noqa: E402
from sc2bench_amd.resnet import resnet50  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--bs', type=int, default=256)
ap.add_argument('--policy', default='')
args = ap.parse_args()
if args.policy:
    hip.configure(**{k: int(v) for k, v in (kv.split('=') for kv in args.policy.split(','))})
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
student = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False).to(dev)
teacher = resnet50().to(dev)
stage = T.DistillationStage(teacher, student, B.STAGE1, dev, head_dtype=torch.bfloat16)
x = B.synthetic_batch(args.bs, dev, seed=0)
phases = {}


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def step(record):
    marks = [ev()]
    tb = x.to(stage.head_dtype).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t_out = stage._run_sequential(stage._teacher_sequence(), stage.t_hooks, tb, with_grad=False)
    t_io = stage.t_hooks.pop()
    t_io['.'] = {'output': t_out}
    marks.append(ev())
    s_out = stage._student_forward(x)
    s_io = stage.s_hooks.pop()
    s_io['.'] = {'output': s_out}
    marks.append(ev())
    loss = stage.criterion(s_io, t_io, None)
    marks.append(ev())
    stage.post_forward_process(loss, bottleneck_updated=False)
    marks.append(ev())
    if record:
        torch.cuda.synchronize()
        for name, a, b in zip(('teacher forward', 'student forward', 'criterion', 'backward + step'), marks, marks[1:]):
            phases.setdefault(name, []).append(a.elapsed_time(b))


for _ in range(3):
    step(False)
torch.cuda.synchronize()
for _ in range(args.steps):
    step(True)
print('phases (ms per step, no per-launch events):')
for k, v in phases.items():
    print('  {:<18}{:8.3f}'.format(k, sum(v) / len(v)))
print('  {:<18}{:8.3f}'.format('sum', sum(sum(v) / len(v) for v in phases.values())))
with hip.KernelTimer() as kt:
    for _ in range(args.steps):
        step(False)
torch.cuda.synchronize()
rows = sorted(((n * ms / args.steps, tag, n / args.steps, ms) for tag, (n, ms) in kt.summary().items()), reverse=True)
tot = sum(r[0] for r in rows)
print('tagged launches: {:.3f} ms per step'.format(tot))
groups = {}
for per_step, tag, n, ms in rows:
    t = tag
    for pre in ('layer1', 'layer2', 'layer3', 'layer4'):
        if tag.startswith(pre):
            t = pre + ('.dgrad' if tag.endswith('dgrad') else '')
    groups.setdefault(t, [0.0, 0.0])
    groups[t][0] += per_step
    groups[t][1] += n
for t, (ms, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print('  {:<40}{:8.3f} ms  {:6.1f} launches'.format(t, ms, n))

# the individual launches of the coarse tags, in issue order, for the last profiled step
per = len(kt.records) // args.steps
for tag, e0, e1 in kt.records[-per:]:
    if tag in ('wgrad', 'dgrad', 'gdn.bwd.pre', 'gdn.bwd.post', 'stem') or tag.startswith(('enc.', 'dec.', 'layer1', 'gdn.rows')):
        print('    {:<20}{:8.3f}'.format(tag, e0.elapsed_time(e1)))
