#!/usr/bin/env python
"""Host-side (Python) cost of one bench step: cProfile over N steps with the GPU running asynchronously.
Shows which wrappers dominate the launch path when the step becomes host-bound."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.Pipeline(6).to(dev).train()
frames = [bench.make_cloud(116000, 0, dev)]
params = [p for p in model.parameters() if p.requires_grad]


def step():
    for p in params:
        p.grad = None
    out = model(frames)
    out.sum().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
# host issue time per step (includes the waits inside .item() readbacks) vs wall time per step
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.2f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
