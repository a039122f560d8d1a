#!/usr/bin/env python
"""Host-side (Python) cost of one step of `bench.py --workload fsd | fsdv2`: cProfile over N steps with the GPU running
asynchronously - which wrappers dominate the launch path of a host-bound step.
Usage: host_profile_workload.py fsd|fsdv2 [steps] [cumulative]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as BW  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'fsd'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = BW.WORKLOADS[name]
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = spec['cls']().to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
clouds = [model.make_cloud(spec['points'], 0, dev)]


def step():
    for p in params:
        p.grad = None
    loss, _ = model(clouds, prepared=None)
    loss.backward()


for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue {1e3 * (t1 - t0) / steps:.2f} ms/step, wall {1e3 * (t2 - t0) / steps:.2f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
