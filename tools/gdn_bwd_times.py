"""Development aid: GDN1 backward per layer shape of the bottleneck at bs 256 (the four GDN layers are 9.5 ms of the 39 ms
stage-1 step: norm GEMM + gamma^T GEMM + weight gradient on the round-1 tile kernels, two HBM-bound element-wise passes)."""
import sys,os,torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sc2bench_amd as S
from tools import env_policy  # noqa: E402  (the SC2_* variables of the A/B scripts -> the dispatch policy)
env_policy.apply()
from sc2bench_amd import hip
dev=torch.device('cuda:0')
for C,HW in ((512,56*56),(96,112*112),(256,55*55),(48,56*56)):
    M=256*HW
    x=torch.randn(M,C,device=dev).to(torch.bfloat16); gy=torch.randn_like(x)
    g=S.GDN1(C,inverse=True).to(dev)
    beta,gamma=g.beta_reparam(g.beta).detach(),g.gamma_reparam(g.gamma).detach()
    xn=x.view(256,int(HW**0.5),-1,C) if int(HW**0.5)**2==HW else x.view(256,HW,1,C)
    f=lambda: hip.gdn1_backward(gy.view_as(xn),xn,beta,gamma,True)
    for _ in range(2): f()
    torch.cuda.synchronize()
    with hip.KernelTimer(lambda t: True) as kt:
        for _ in range(5): f()
        torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print(C,HW,'gdn1_backward total %.3f ms'%(e0.elapsed_time(e1)/5), {k:round(v[1],3) for k,v in kt.summary().items()})
