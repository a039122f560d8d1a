#!/usr/bin/env python
"""Runs only the SRA attention-core kernels on the bench workload (for rocprofv3 --pmc passes).
Usage: sra_only.py [iters] [sorted]   ('sorted': token rows laid out in window order, i.e. tok = arange)"""
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
if len(sys.argv) > 2 and sys.argv[2] == 'sorted':
    assert plan.n_tokens == m
    plan = K.WindowPlan(torch.arange(m, dtype=torch.int32, device=mb.DEV), plan.winoff, plan.n_windows, m,
                        plan.max_tokens)
qkv = torch.randn(m, 384, device=mb.DEV)
do = torch.randn(m, 128, device=mb.DEV)
dqkv = torch.empty_like(qkv)
q, k, v = qkv[:, :128], qkv[:, 128:256], qkv[:, 256:]
for _ in range(iters):
    o, lse = K._sra_fwd(q, k, v, plan, 8, 0.25, 0)
    K._sra_bwd(q, k, v, o, lse, do, plan, 8, 0.25, 0, dqkv[:, :128], dqkv[:, 128:256], dqkv[:, 256:])
# scaled cosine attention inside the same kernels (normalisation in the load prologue, per-head 1 / clamp(tau) from device memory)
hs = torch.full((8,), 4.0, device=mb.DEV)
if K.cosine_kernels_ok(plan, 8):
    for _ in range(iters):
        oc, lsec = K._sra_cos_fwd(q, k, v, plan, 8, hs)
        K._sra_cos_bwd(q, k, v, oc, lsec, do, plan, 8, hs, dqkv[:, :128], dqkv[:, 128:256], dqkv[:, 256:])
# the reduced-precision kernels on the same plan (bf16 storage)
from sst_amd import bf16  # noqa: E402
qb, kb, vb, dob = (t.to(torch.bfloat16).contiguous() for t in (q, k, v, do))
dqb, dkb, dvb = (torch.empty_like(qb) for _ in range(3))
for _ in range(iters):
    ob, lseb = bf16.sra_fwd(qb, kb, vb, plan, 8, 0.25)
    bf16.sra_bwd(qb, kb, vb, ob, lseb, dob, plan, 8, 0.25, dqb, dkb, dvb)
torch.cuda.synchronize()
print('tokens', m, 'windows', plan.n_windows)
