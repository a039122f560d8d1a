#!/usr/bin/env python
"""Runs only the SRA attention-core kernels on the bench workload (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microbench as mb  # noqa: E402
from sst_amd import kernels as K  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
_, _, info = mb.frame_plan()
m = info['voxel_feats'].size(0)
plan = info['sra_plan_shift0']
qk = torch.randn(m, 256, device=mb.DEV)
v = torch.randn(m, 128, device=mb.DEV)
do = torch.randn(m, 128, device=mb.DEV)
dqk = torch.empty_like(qk)
dv = torch.empty_like(v)
for _ in range(iters):
    o, lse = K._sra_fwd(qk[:, :128], qk[:, 128:], v, plan, 8, 0.25, 0)
    K._sra_bwd(qk[:, :128], qk[:, 128:], v, o, lse, do, plan, 8, 0.25, 0, dqk[:, :128], dqk[:, 128:], dv)
torch.cuda.synchronize()
print('tokens', m, 'windows', plan.n_windows)
