#!/usr/bin/env python
"""recover_bev + first attached convolution (3 x 3, dilation 2, 128 -> 128, fp32) forward + backward on a 468 x 468 canvas:
the dense path (canvas -> MIOpen convolution) against sparse_first_conv (sst_amd/backbones.py) at several occupancies,
uniformly random voxels and voxels clustered like a LiDAR sweep (rings).  Usage: python tools/bev_first_conv_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sst_amd.backbones import recover_bev, sparse_first_conv  # noqa: E402

DEV = torch.device('cuda:0')


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    torch.backends.cudnn.benchmark = True
    ny = nx = 468
    conv = torch.nn.Conv2d(128, 128, 3, dilation=2, padding=2, bias=False).to(DEV)
    rng = np.random.default_rng(0)
    gout = torch.randn(1, 128, ny, nx, device=DEV)
    for kind, m in (('uniform', 90107), ('uniform', 45000), ('uniform', 18000), ('rings', 18000), ('rings', 45000)):
        if kind == 'uniform':
            cells = rng.choice(ny * nx, size=m, replace=False)
        else:   # points on 64 rings around the centre, thinned to m distinct cells
            r = rng.choice(np.linspace(8, 230, 64), size=8 * m)
            a = rng.random(8 * m) * 2 * np.pi
            cy, cx = (ny / 2 + r * np.sin(a)).astype(np.int64), (nx / 2 + r * np.cos(a)).astype(np.int64)
            cells = np.unique(np.clip(cy, 0, ny - 1) * nx + np.clip(cx, 0, nx - 1))
            cells = rng.permutation(cells)[:m]
        m = len(cells)
        coors = torch.from_numpy(np.stack([np.zeros(m, np.int64), np.zeros(m, np.int64), cells // nx, cells % nx], 1)).to(DEV)
        feats = torch.randn(m, 128, device=DEV, requires_grad=True)

        def dense():
            conv.zero_grad()
            feats.grad = None
            (conv(recover_bev(feats, coors, 1, (ny, nx))) * gout).sum().backward()

        def sparse():
            conv.zero_grad()
            feats.grad = None
            (sparse_first_conv(feats, coors, 1, (ny, nx), conv) * gout).sum().backward()

        td, ts = timeit(dense), timeit(sparse)
        print(f'{kind:8s} {m:6d} voxels = {100.0 * m / (ny * nx):5.1f} % of the cells: dense {td:6.3f} ms, sparse {ts:6.3f} ms')


if __name__ == '__main__':
    main()
