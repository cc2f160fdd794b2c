"""Development aid: where the host time of one pipelined eval step goes (cProfile over the launch-issuing code)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
dev = torch.device('cuda:0')
model = B.build_model(dev)
x = B.synthetic_batch(256, dev)
with torch.no_grad():
    for _ in range(3):
        sym, hw = model.stage_front(x); dec, nb, st = model.stage_coder(sym, hw); model.stage_back(dec, hw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        sym, hw = model.stage_front(x); dec, nb, st = model.stage_coder(sym, hw); out = model.stage_back(dec, hw)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
print(s.getvalue()[:6000])
