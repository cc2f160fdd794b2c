"""Checks every data-gradient launch of frozen.FrozenStack against torch autograd (f32) on the device."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sc2bench_amd as S  # noqa: E402
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd.frozen import FrozenStack  # noqa: E402
from sc2bench_amd.resnet import resnet50  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = resnet50().eval()
for p in net.parameters():
    p.requires_grad_(False)
net.to(dev)
for name, cin, hw in (('layer2', 256, 56), ('layer3', 512, 28), ('layer4', 1024, 14)):
    stack = FrozenStack(name, getattr(net, name))
    dgs = stack._dg()
    for bi, (blk, dg) in enumerate(zip(stack.blocks, dgs)):
        if bi > 1:
            break
        for ci, (c, d) in enumerate(zip(blk, dg)):
            if c is None:
                continue
            in_hw = hw if (bi == 0 and ci in (0, 1, 3)) else hw // 2
            if ci == 1 and bi == 0:
                in_hw = hw
            cin_c = c.w_folded.shape[1]
            x = torch.randn(2, cin_c, in_hw, in_hw, device=dev).to(torch.bfloat16).float().requires_grad_(True)
            y = F.conv2d(x, c.w_folded.to(torch.bfloat16).float(), None, c.stride, c.pad)
            g = torch.randn_like(y).to(torch.bfloat16)
            y.backward(g.float())
            got = d(g.permute(0, 2, 3, 1).contiguous(), (in_hw, in_hw)).permute(0, 3, 1, 2).float()
            rel = ((got - x.grad).norm() / x.grad.norm()).item()
            print(name, bi, 'c1 c2 c3 ds'.split()[ci], 'k', c.k, 's', c.stride, tuple(x.shape), '->', tuple(y.shape),
                  'via', 'conv' if d.as_conv is not None else 'dgrad', 'rel', round(rel, 5))
