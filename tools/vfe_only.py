#!/usr/bin/env python
"""Runs only voxelize + DynamicVFE forward/backward of the bench workload (for rocprofv3 kernel statistics)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.Pipeline(6).to(dev).train()
frames = [bench.make_cloud(116000, 0, dev)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for it in range(iters):
    points, coors = model.voxel_layer.voxelize_batch(frames)
    vf, vc = model.voxel_encoder(points, coors)
    vf.sum().backward()
torch.cuda.synchronize()
print('voxels', vf.shape)
