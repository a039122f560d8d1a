#!/usr/bin/env python
"""Segmented reduction at FSD sizes, stand-alone (GPU box): Zipf-sized groups as FSD's clusters have them, the balanced
tile kernel (seg_tiles_k) beside the per-group kernels, HIP events around back-to-back launches; also the loop rocprofv3
--kernel-trace --stats / --pmc is pointed at.  Usage: python tools/seg_reduce_only.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sst_amd import kernels as K  # noqa: E402

DEV = torch.device('cuda:0')


def grouping(n, k, seed):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, k + 1) ** 1.1
    ids = rng.choice(k, size=n, p=w / w.sum())
    ids[:k] = np.arange(k)
    coors = torch.from_numpy(np.stack([np.zeros(n, np.int64), ids // 64, ids % 64], 1)).to(DEV)
    return K.unique_rows(coors)


def timed(fn, reps):
    """kernel-bound HIP events (hipExtLaunchKernelGGL start / stop) on every launch: a Python call costs more than these
    kernels run, events around a loop of calls would time the host"""
    from sst_amd import _lib
    lib = _lib.load()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        ke = K._KernelEvents(lib)
        lib.sst_segment_reduce_profile_next(ke.start, ke.stop)
        fn()
        lib.sst_segment_reduce_profile_next(None, None)
        torch.cuda.synchronize()
        ts.append(ke.elapsed_time() * 1e3)
    return float(np.median(ts))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    for n, k, c, mode in ((18443, 1554, 128, 'max'), (50000, 1554, 128, 'max'), (18443, 1554, 3, 'mean'),
                          (160000, 6000, 128, 'max'), (300000, 20000, 64, 'max')):
        plan = grouping(n, k, n)
        sizes = plan.counts().cpu().numpy()
        x = torch.randn(n, c, device=DEV)
        by = (4 * c + 4) * n + 4 * c * k
        res = {}
        for flag in ('1', '0'):
            os.environ['SST_SEG_LONG'] = flag
            with torch.no_grad():
                res[flag] = timed(lambda: K.segment_reduce(x, plan, mode), reps)
        print(f'{mode} n={n} groups={k} (largest {sizes.max()}, median {int(np.median(sizes))}) c={c}: {by / 1e6:.1f} MB; '
              f'tile kernel {res["1"]:.1f} us = {by / res["1"] / 1e6:.2f} TB/s ({100 * by / res["1"] / 1e6 / 8:.1f} % of 8 TB/s); '
              f'per-group kernels {res["0"]:.1f} us = {by / res["0"] / 1e6:.2f} TB/s')


if __name__ == '__main__':
    main()
