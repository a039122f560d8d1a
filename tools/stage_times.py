#!/usr/bin/env python
"""Stage-level timing of one bench step (HIP events, median of N): voxelize / VFE / input layer / backbone fwd / bwd."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
import torch.cuda.tunable as tunable
tunable.enable(True)
torch.manual_seed(0)
model = bench.Pipeline(6).to(dev).train()
frames = [bench.make_cloud(116000, 0, dev)]
names = ['voxelize', 'vfe', 'input_layer', 'backbone_fwd', 'backward']
acc = {n: [] for n in names}


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for it in range(25):
    for p in model.parameters():
        p.grad = None
    e0 = ev()
    points, coors = model.voxel_layer.voxelize_batch(frames)
    e1 = ev()
    vf, vc = model.voxel_encoder(points, coors)
    e2 = ev()
    info = model.middle_encoder(vf, vc, 1)
    e3 = ev()
    out = model.backbone(info)[0]['voxel_feats']
    e4 = ev()
    out.sum().backward()
    e5 = ev()
    torch.cuda.synchronize()
    if it >= 5:
        for n, a, b in zip(names, (e0, e1, e2, e3, e4), (e1, e2, e3, e4, e5)):
            acc[n].append(a.elapsed_time(b))
tot = 0
for n in names:
    m = float(np.median(acc[n]))
    tot += m
    print(f'{n:14s} {m:8.3f} ms')
print(f'{"total":14s} {tot:8.3f} ms')
