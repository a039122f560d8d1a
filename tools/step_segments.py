#!/usr/bin/env python
"""Where the device time of one SST step goes, WITHOUT a profiler attached (rocprofv3 slows the host down, and the gaps it
shows at the head of a step are partly its own): events on the stream at the boundaries of a step

    S0 step begins | S1 voxel_info done (index plan consumed, voxel encoder, gather) | S2 forward of the blocks done |
    S3 backward done | S4 next plan queued (prefetch modes)

for three ways of building the index plan: inside the step (`none`), ahead on the same stream behind the backward pass
(`same_stream`), ahead on the planner's own stream beside the backward pass (`overlap`: FramePlanner.build_overlapped), ahead on the same stream between this step's forward and backward pass (`mid_step`).
Prints one JSON object: per mode the median step (wall clock over the loop / steps, and event to event) and the median
segments in ms.  Usage: step_segments.py [steps] [points] [lidar]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
points = int(sys.argv[2]) if len(sys.argv) > 2 else 116000
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.Pipeline(6).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
frames = [bench.make_lidar_cloud(2000, dev)] if 'lidar' in sys.argv else [bench.make_cloud(points, 1000, dev)]
seed = {}
res = {'points': int(frames[0].size(0))}


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def run(mode, n, marks):
    ahead = []
    for i in range(n):
        row = [ev()]
        for p in params:
            p.grad = None
        plan = ahead.pop() if ahead else None
        info = model.voxel_info(frames, plan)
        row.append(ev())
        out = model.backbone.forward_voxels(info)
        row.append(ev())
        g = seed.get(out.shape)
        if g is None:
            g = seed[out.shape] = torch.randn(out.shape, device=dev)
        if mode == 'mid_step':       # between the forward and the backward pass, same stream (bench_workloads.py does this for FSD)
            ahead.append(model.prepare(frames))
        out.backward(g)
        del info, out, plan
        row.append(ev())
        if mode in ('same_stream', 'overlap'):
            ahead.append(model.prepare(frames, overlap=(mode == 'overlap')))
        row.append(ev())
        if marks is not None:
            marks.append(row)
    ahead.clear()


import gc
for mode in ('none', 'same_stream', 'overlap', 'mid_step', 'none', 'overlap', 'mid_step'):
    run(mode, 6, None)
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    run(mode, steps, marks)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    seg = {k: [] for k in ('front', 'forward', 'backward', 'plan_tail', 'step')}
    for a, b in zip(marks[:-1], marks[1:]):
        seg['front'].append(a[0].elapsed_time(a[1]))
        seg['forward'].append(a[1].elapsed_time(a[2]))
        seg['backward'].append(a[2].elapsed_time(a[3]))
        seg['plan_tail'].append(a[3].elapsed_time(a[4]) + a[4].elapsed_time(b[0]))
        seg['step'].append(a[0].elapsed_time(b[0]))
    key = mode if mode not in res else mode + '_again'
    res[key] = {'ms_per_step_wall': round(wall, 3)}
    res[key].update({k: round(statistics.median(v), 3) for k, v in seg.items()})
    res[key]['voxels'] = int(model.last_voxel_coors.size(0)) if hasattr(model, 'last_voxel_coors') else None
print(json.dumps(res))
