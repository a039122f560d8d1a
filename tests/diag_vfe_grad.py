#!/usr/bin/env python
"""Developer diagnostic (GPU box): where do the DynamicVFE parameter gradients of the GPU pipeline and of the CPU port
of the reference flow (oracle/cpu_pipeline.py) part?  Same frames as tests/test_gpu_end_to_end.py (crowded cloud), same
weights; prints forward / gradient differences of every intermediate tensor of the VFE, stage by stage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import bench  # noqa: E402
from oracle.cpu_pipeline import CpuSSTBackbone, load_pipeline_weights  # noqa: E402

DEV = 'cuda:0'
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
torch.manual_seed(0)
gpu = bench.Pipeline(1).to(DEV).train()
gpu.middle_encoder.shuffle_voxels = False
cpu = load_pipeline_weights(CpuSSTBackbone(bench.VOXEL_SIZE, bench.PC_RANGE, bench.DROP_TRAIN, num_blocks=1).train(), gpu)
g = torch.Generator().manual_seed(1)
pts = bench.make_cloud(n_points, 5, 'cpu')
dense = torch.rand(n_points // 3, 3, generator=g) * torch.tensor([7.0, 7.0, 6.0]) + torch.tensor([10.0, 10.0, -2.0])
frames = [torch.cat([pts, dense]), bench.make_cloud(n_points // 2, 6, 'cpu')]

# ---- GPU VFE with retained intermediates -----------------------------------------------------------------
points, coors = gpu.voxel_layer.voxelize_batch([f.to(DEV) for f in frames])
sp = gpu.voxel_encoder.scatter_plan(coors)
kept = {}
for i, layer in enumerate(gpu.voxel_encoder.vfe_layers):
    def mk(i, fwd):
        def f(x):
            kept[f'in{i}'] = x
            if x.requires_grad:
                x.retain_grad()
            y = fwd(x)
            y.retain_grad()
            kept[f'pf{i}'] = y
            return y
        return f
    layer.forward = mk(i, layer.forward)
vf_g, vc_g = gpu.voxel_encoder(points, coors, scatter_plan=sp)
vf_g.retain_grad()

# ---- CPU VFE, same structure ------------------------------------------------------------------------------
import numpy as np  # noqa: E402
from oracle import voxel_oracle  # noqa: E402
import torch.nn.functional as F  # noqa: E402
cc = [np.pad(voxel_oracle.dynamic_voxelize(p.numpy(), bench.VOXEL_SIZE, bench.PC_RANGE), ((0, 0), (1, 0)),
             constant_values=b) for b, p in enumerate(frames)]
cc = torch.from_numpy(np.concatenate(cc))
pp = torch.cat(frames)
vfe = cpu.vfe
mean, vcoors = voxel_oracle.DynamicScatterOracle(None, None, True)(pp, cc)
key = lambda c: ((c[:, 0].long() * 4 + c[:, 1].long()) * 4096 + c[:, 2].long()) * 4096 + c[:, 3].long()
vkey = key(vcoors)
pos = torch.searchsorted(vkey, key(cc)).clamp(max=vkey.numel() - 1)
inv = torch.where(vkey[pos] == key(cc), pos, torch.zeros_like(pos))
f_cluster = pp[:, :3] - mean[inv][:, :3]
f_center = torch.stack([pp[:, 0] - (cc[:, 3].float() * vfe.vx + vfe.x_offset),
                        pp[:, 1] - (cc[:, 2].float() * vfe.vy + vfe.y_offset),
                        pp[:, 2] - (cc[:, 1].float() * vfe.vz + vfe.z_offset)], 1)
feats = torch.cat([pp, f_cluster, f_center], 1)
ck = {}
smax = voxel_oracle.DynamicScatterOracle(None, None, False)
for i, (lin, norm) in enumerate(zip(vfe.linears, vfe.norms)):
    ck[f'in{i}'] = feats
    if feats.requires_grad:
        feats.retain_grad()
    pf = F.relu(norm(lin(feats)))
    pf.retain_grad()
    ck[f'pf{i}'] = pf
    vf, vcoors = smax(pf, cc)
    if i != len(vfe.linears) - 1:
        feats = torch.cat([pf, vf[inv]], 1)
vf.retain_grad()
assert torch.equal(vc_g.cpu().long(), vcoors.long())

gout = torch.randn(vf.shape, generator=g)
(vf_g * gout.to(DEV)).sum().backward()
(vf * gout).sum().backward()


def rel(a, b):
    return float((a.detach().cpu() - b.detach()).abs().max()) / max(1e-12, float(b.detach().abs().max()))


print('points', pp.size(0), 'voxels', vf.size(0))
print('fwd voxel feats        ', rel(vf_g, vf))
for i in (1, 0):
    print(f'fwd point feats {i}      ', rel(kept[f'pf{i}'], ck[f'pf{i}']))
    print(f'grad point feats {i}     ', rel(kept[f'pf{i}'].grad, ck[f'pf{i}'].grad),
          ' rows differing > 1e-4 of max:',
          int(((kept[f'pf{i}'].grad.cpu() - ck[f'pf{i}'].grad).abs().max(1).values
               > 1e-4 * ck[f'pf{i}'].grad.abs().max()).sum()))
    if kept[f'in{i}'].grad is not None:
        print(f'grad layer input {i}     ', rel(kept[f'in{i}'].grad, ck[f'in{i}'].grad))
    wg, wc = gpu.voxel_encoder.vfe_layers[i].linear.weight.grad, vfe.linears[i].weight.grad
    print(f'grad linear weight {i}   ', rel(wg, wc))
    print(f'grad norm weight {i}     ', rel(gpu.voxel_encoder.vfe_layers[i].norm.weight.grad, vfe.norms[i].weight.grad))
    # the weight gradient recomputed in float64 from the GPU's own tensors: dW = d(pre)^T x cannot be formed without
    # d(pre); use autograd on the CPU with the GPU's point-feature gradient instead
    x = ck[f'in{i}'].detach().double()
    lin = torch.nn.Linear(x.size(1), wc.size(0), bias=False).double()
    lin.weight.data.copy_(vfe.linears[i].weight.detach().double())
    bn = torch.nn.BatchNorm1d(wc.size(0), eps=1e-3, momentum=0.01).double().train()
    bn.weight.data.copy_(vfe.norms[i].weight.detach().double())
    bn.bias.data.copy_(vfe.norms[i].bias.detach().double())
    y = F.relu(bn(lin(x)))
    (y * kept[f'pf{i}'].grad.cpu().double()).sum().backward()
    print(f'  float64 dW from the GPU point-feature gradient vs GPU dW {rel(wg.double(), lin.weight.grad):.2e}, '
          f'vs CPU dW {rel(wc.double(), lin.weight.grad):.2e}')
