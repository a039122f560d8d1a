#!/usr/bin/env python
"""Encoder stack with batch-norm layers (configs/fsd/fsd_waymoD1_1x_sst_encoder.py:70: layer_cfg use_bn=True, cosine=True,
tau_min=0.01) on the bench frame (90 k voxels): forward + backward of the SSTv2 blocks on the fused chain and module by module,
beside the same stack with LayerNorm.  Usage (GPU box): python tools/bn_layers_bench.py [blocks]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microbench as mb  # noqa: E402
import sst_amd  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
_, _, info = mb.frame_plan()
m = info['voxel_feats'].size(0)
g = torch.randn(m, 128, device=mb.DEV)
res = {'voxels': m, 'blocks': blocks}
for tag, cfg in (('layer_norm_cosine', dict(cosine=True, tau_min=0.01)), ('batch_norm_cosine', dict(use_bn=True, cosine=True, tau_min=0.01))):
    torch.manual_seed(0)
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128] * blocks, nhead=[8] * blocks, num_blocks=blocks,
                                      dim_feedforward=[256] * blocks, output_shape=[468, 468], num_attached_conv=0, to_bev=False,
                                      debug=False, layer_cfg=cfg)).to(mb.DEV).train()
    for fused in (True, False):
        net.set_fused(fused)
        x = info['voxel_feats'].detach().clone().requires_grad_(True)

        def step():
            for p in net.parameters():
                p.grad = None
            vi = dict(info)
            vi['voxel_feats'] = x
            out = net(vi)[0]['voxel_feats']
            out.backward(g)
        med, mn = mb.timeit(step, iters=10, warmup=3)
        res[f'{tag}.{"fused" if fused else "modular"}_ms'] = round(med, 3)
print(json.dumps(res))
