"""TEST INFRASTRUCTURE ONLY — the whole SST backbone hot path on CPU, following the reference's algorithm
(padded per-level windows + nn.MultiheadAttention with a key-padding mask, atomics-free DynamicScatter
restatement), used (1) as bench.py's ``cpu_baseline`` ("port" of the reference path, timed on the host
cores) and (2) as an end-to-end cross-check of the GPU pipeline.

Reference path restated: DynamicVoxelNet.extract_feat (mmdet3d/models/detectors/dynamic_voxelnet.py:38-47):
voxelize -> DynamicVFE (voxel_encoders/voxel_encoder.py:230-298) -> SSTInputLayerV2
(middle_encoders/sst_input_layer_v2.py:80-126) -> SSTv2 blocks (backbones/sst_v2.py:115-154,
sst/sst_basic_block_v2.py:41-126).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sst_oracle, voxel_oracle


class CpuDynamicVFE(nn.Module):

    def __init__(self, voxel_size, point_cloud_range, feat_channels=(64, 128), in_channels=3):
        super().__init__()
        self.vx, self.vy, self.vz = voxel_size
        r = point_cloud_range
        self.x_offset, self.y_offset, self.z_offset = self.vx / 2 + r[0], self.vy / 2 + r[1], self.vz / 2 + r[2]
        chans = [in_channels + 6] + list(feat_channels)
        self.linears = nn.ModuleList()
        self.norms = nn.ModuleList()
        for i in range(len(chans) - 1):
            cin = chans[i] * (2 if i > 0 else 1)
            self.linears.append(nn.Linear(cin, chans[i + 1], bias=False))
            self.norms.append(nn.BatchNorm1d(chans[i + 1], eps=1e-3, momentum=0.01))

    def forward(self, features, coors):
        mean, vcoors = voxel_oracle.DynamicScatterOracle(None, None, True)(features, coors)
        # point -> voxel index (canvas default 0, voxel_encoder.py:185-225)
        key = lambda c: ((c[:, 0].long() * 4 + c[:, 1].long()) * 4096 + c[:, 2].long()) * 4096 + c[:, 3].long()
        vkey = key(vcoors)
        pos = torch.searchsorted(vkey, key(coors)).clamp(max=vkey.numel() - 1)
        inv = torch.where(vkey[pos] == key(coors), pos, torch.zeros_like(pos))
        f_cluster = features[:, :3] - mean[inv][:, :3]
        f_center = torch.stack([features[:, 0] - (coors[:, 3].float() * self.vx + self.x_offset),
                                features[:, 1] - (coors[:, 2].float() * self.vy + self.y_offset),
                                features[:, 2] - (coors[:, 1].float() * self.vz + self.z_offset)], 1)
        feats = torch.cat([features, f_cluster, f_center], 1)
        scatter_max = voxel_oracle.DynamicScatterOracle(None, None, False)
        keep = getattr(self, 'keep_point_feats', None)   # tests: per-point features of every layer (with gradients)
        for i, (lin, norm) in enumerate(zip(self.linears, self.norms)):
            pf = F.relu(norm(lin(feats)))
            if keep is not None:
                pf.retain_grad()
                keep.append(pf)
            vf, vcoors = scatter_max(pf, coors)
            if i != len(self.linears) - 1:
                feats = torch.cat([pf, vf[inv]], 1)
        return vf, vcoors


class CpuEncoderLayer(nn.Module):
    """Post-norm EncoderLayer on padded windows, exactly the reference's data flow."""

    def __init__(self, d_model=128, nhead=8, ffn=256):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=0.0)
        self.linear1 = nn.Linear(d_model, ffn)
        self.linear2 = nn.Linear(ffn, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, levels):
        out = torch.zeros_like(src)
        for (vox_idx, slot_idx, n_win, t) in levels:
            c = src.size(1)
            feat3d = src.new_zeros((n_win * t, c))
            pos3d = src.new_zeros((n_win * t, c))
            mask = torch.ones(n_win * t, dtype=torch.bool)
            feat3d[slot_idx] = src[vox_idx]
            pos3d[slot_idx] = pos[vox_idx].to(src.dtype)
            mask[slot_idx] = False
            feat3d = feat3d.view(n_win, t, c).permute(1, 0, 2)
            qk = feat3d + pos3d.view(n_win, t, c).permute(1, 0, 2)
            o, _ = self.self_attn(qk, qk, value=feat3d, key_padding_mask=mask.view(n_win, t))
            out[vox_idx] = o.permute(1, 0, 2).reshape(-1, c)[slot_idx]
        src = self.norm1(src + out)
        src2 = self.linear2(F.gelu(self.linear1(src)))
        return self.norm2(src + src2)


class CpuSSTBackbone(nn.Module):

    def __init__(self, voxel_size, point_cloud_range, drop_info, num_blocks=6, d_model=128, nhead=8, ffn=256,
                 window_shape=(12, 12, 1), sparse_shape=(468, 468, 1)):
        super().__init__()
        self.voxel_size, self.pc_range, self.drop_info = voxel_size, point_cloud_range, drop_info
        self.window_shape, self.sparse_shape, self.d_model = window_shape, sparse_shape, d_model
        self.vfe = CpuDynamicVFE(voxel_size, point_cloud_range, (64, d_model))
        self.layers = nn.ModuleList([CpuEncoderLayer(d_model, nhead, ffn) for _ in range(2 * num_blocks)])

    def plan(self, vcoors):
        c = vcoors.numpy().astype(np.int64)
        w0, c0 = sst_oracle.window_coors(c, self.sparse_shape, self.window_shape, False)
        w1, c1 = sst_oracle.window_coors(c, self.sparse_shape, self.window_shape, True)
        rb = sst_oracle.region_batching(w0, w1, self.drop_info)
        keep = rb['keep_idx']
        shifts = []
        for s, ciw in enumerate((c0, c1)):
            pos = torch.from_numpy(sst_oracle.pos_embed(ciw[keep], self.window_shape, self.d_model))
            levels = []
            for dl in self.drop_info:
                msk = rb[f'level{s}'] == dl
                if not msk.any():
                    continue
                t = self.drop_info[dl]['max_tokens']
                slots = rb[f'flat2win{s}'][msk]
                levels.append((torch.from_numpy(np.nonzero(msk)[0]), torch.from_numpy(slots),
                               int(slots.max() // t + 1), t))
            shifts.append((pos, levels))
        return torch.from_numpy(keep), shifts

    def forward(self, points_list):
        coors = [np.pad(voxel_oracle.dynamic_voxelize(p.float().numpy(), self.voxel_size, self.pc_range), ((0, 0), (1, 0)),
                        constant_values=b) for b, p in enumerate(points_list)]
        coors = torch.from_numpy(np.concatenate(coors))
        pts = torch.cat(points_list)
        vf, vc = self.vfe(pts, coors)
        keep, shifts = self.plan(vc)
        self.last_voxel_coors = vc[keep]      # (b, z, y, x) of the rows of the output, sorted-unique order
        self.last_all_voxel_coors = vc        # ... of every voxel before the drop
        x = vf[keep]
        for i, layer in enumerate(self.layers):
            pos, levels = shifts[i % 2]
            x = layer(x, pos, levels)
        return x


def load_pipeline_weights(cpu, gpu_pipeline):
    """Copy the parameters of a bench.Pipeline (reference parameter names: ``vfe_layers.{i}.linear / norm``,
    ``block_list.{i}.encoder_list.{0,1}.*``) into a CpuSSTBackbone, so both run the same network."""
    with torch.no_grad():
        for i, layer in enumerate(gpu_pipeline.voxel_encoder.vfe_layers):
            cpu.vfe.linears[i].weight.copy_(layer.linear.weight.cpu())
            cpu.vfe.norms[i].weight.copy_(layer.norm.weight.cpu())
            cpu.vfe.norms[i].bias.copy_(layer.norm.bias.cpu())
        k = 0
        for block in gpu_pipeline.backbone.block_list:
            for enc in block.encoder_list:
                dst = cpu.layers[k]
                k += 1
                dst.self_attn.load_state_dict({n: p.cpu() for n, p in enc.win_attn.self_attn.state_dict().items()})
                for name in ('linear1', 'linear2', 'norm1', 'norm2'):
                    getattr(dst, name).load_state_dict({n: p.cpu() for n, p in getattr(enc, name).state_dict().items()})
        assert k == len(cpu.layers)
    return cpu


def voxel_sort_key(coors):
    """(b, z, y, x) -> one int64 key whose order is the sorted-unique voxel order of the reference"""
    c = coors.long()
    return ((c[:, 0] * 64 + c[:, 1]) * 4096 + c[:, 2]) * 4096 + c[:, 3]
