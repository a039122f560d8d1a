#!/usr/bin/env python
"""Runs only the fp32 LDS-resident linear kernels (csrc/dense_f32.hip) at the bench size, for rocprofv3 --pmc passes.
Usage: lds_linear_only.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sst_amd import dense as D  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = 90107
g = torch.Generator().manual_seed(0)
for k, n in ((128, 128), (128, 256), (256, 128)):
    x = torch.randn(m, k, generator=g).cuda()
    w = torch.randn(n, k, generator=g).cuda() / k ** 0.5
    b = torch.randn(n, generator=g).cuda()
    for _ in range(iters):
        D.lds_linear(x, w, b)
torch.cuda.synchronize()
