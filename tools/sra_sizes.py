#!/usr/bin/env python
"""SRA attention-core kernels alone (GPU box) on the window distributions of the bench's frames: the uniform cloud (90 k voxels,
~60 tokens per window) and the LiDAR-like cloud (18 k voxels per frame, 23 tokens per window, a few windows at the 100-token cap)
at 1 / 2 / 4 frames per launch.  Per variant (fp32 standard, fp32 cosine, bf16): median launch time over interleaved rounds
(several launches between two events: the kernels are shorter than an event pair) and the fraction of 8 TB/s its ALGORITHMIC
bytes reach (2 056 B per token forward, 4 104 backward; half of that in the bf16 mode).
Usage: python tools/sra_sizes.py [uniform|lidar|all] [frames ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import microbench as mb  # noqa: E402
import sst_amd  # noqa: E402
from sst_amd import bf16, kernels as K  # noqa: E402

DEV = mb.DEV


def plans(cloud, frames):
    pts = [bench.make_lidar_cloud(2000 + i, DEV) if cloud == 'lidar' else bench.make_cloud(116000, i, DEV) for i in range(frames)]
    vox = sst_amd.Voxelization(bench.VOXEL_SIZE, bench.PC_RANGE, -1, (-1, -1))
    _, coors = vox.voxelize_batch(pts)
    sp = sst_amd.build_scatter_plan(coors, grid_zyx=[1, 468, 468])
    layer = sst_amd.SSTInputLayerV2((bench.DROP_TRAIN, bench.DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, mute=True,
                                    reference_outputs=False, debug=False)
    layer.train(True)
    info = layer(torch.randn(sp.num_voxels, 128, device=DEV), sp.voxel_coors, frames)
    return info


def measure(cloud, frames, reps=8):
    info = plans(cloud, frames)
    m = info['voxel_feats'].size(0)
    out = {'cloud': cloud, 'frames': frames, 'tokens': m}
    for s in range(2):
        plan = info[f'sra_plan_shift{s}']
        sizes = np.diff(plan.winoff[:plan.n_windows + 1].cpu().numpy())
        qkv = torch.randn(m, 384, device=DEV)
        do = torch.randn(m, 128, device=DEV)
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :128], qkv[:, 128:256], qkv[:, 256:]
        hs = torch.full((8,), 4.0, device=DEV)
        row = {'windows': int(plan.n_windows), 'tokens_mean': round(float(sizes.mean()), 1), 'tokens_max': int(sizes.max()),
               'tiles_hist': np.bincount((sizes + 15) // 16).tolist()}
        o, lse = K._sra_fwd(q, k, v, plan, 8, 0.25, 0)
        fb, bb = bench.SRA_BYTES_PER_TOKEN * m, bench.SRA_BWD_BYTES_PER_TOKEN * m

        def rec(tag, fn, nbytes):
            med, mn = mb.timeit(fn, iters=20, warmup=3, reps=reps)
            row[tag] = {'us': round(med * 1e3, 2), 'min_us': round(mn * 1e3, 2), 'frac': round(nbytes / (med * 1e-3) / 8e12, 4)}
        rec('fwd', lambda: K._sra_fwd(q, k, v, plan, 8, 0.25, 0), fb)
        rec('bwd', lambda: K._sra_bwd(q, k, v, o, lse, do, plan, 8, 0.25, 0, dqkv[:, :128], dqkv[:, 128:256], dqkv[:, 256:]), bb)
        if K.cosine_kernels_ok(plan, 8):
            oc, lsec = K._sra_cos_fwd(q, k, v, plan, 8, hs)
            rec('cos_fwd', lambda: K._sra_cos_fwd(q, k, v, plan, 8, hs), fb)
            rec('cos_bwd', lambda: K._sra_cos_bwd(q, k, v, oc, lsec, do, plan, 8, hs, dqkv[:, :128], dqkv[:, 128:256], dqkv[:, 256:]), bb)
        qb, kb, vb, dob = (t.to(torch.bfloat16).contiguous() for t in (q, k, v, do))
        dqb, dkb, dvb = (torch.empty_like(qb) for _ in range(3))
        ob, lseb = bf16.sra_fwd(qb, kb, vb, plan, 8, 0.25)
        rec('bf16_fwd', lambda: bf16.sra_fwd(qb, kb, vb, plan, 8, 0.25), fb // 2)
        rec('bf16_bwd', lambda: bf16.sra_bwd(qb, kb, vb, ob, lseb, dob, plan, 8, 0.25, dqb, dkb, dvb), bb // 2)
        out[f'shift{s}'] = row
    return out


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    fr = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
    for cloud in (('uniform', 'lidar') if which == 'all' else (which,)):
        for f in (fr if cloud == 'lidar' else [f for f in fr if f <= 2]):
            print(json.dumps(measure(cloud, f)), flush=True)
