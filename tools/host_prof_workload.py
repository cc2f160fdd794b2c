"""Development aid: where the HOST spends its time issuing a workload's pipelined steps (cProfile of StagePipeline.run).
    python tools/host_prof_workload.py fp_input [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import sc2bench_amd as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'fp_input'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda:0')
model, x, what, hw, n = B.build_workload(name, dev, 0)
g, c = B.WORKLOAD_PIPELINE[name]
pipe = S.StagePipeline(model, dev, coder_group=g, coder_streams=c)
pipe.run(x, n_steps=8)
pipe.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
pipe.run(x, n_steps=steps)
pr.disable()
t1 = time.perf_counter()
pipe.synchronize()
t2 = time.perf_counter()
print('{}: issue {:.2f} ms/step, done {:.2f} ms/step'.format(name, 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps))
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
