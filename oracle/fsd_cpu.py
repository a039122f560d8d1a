"""TEST INFRASTRUCTURE ONLY — the FSD / FSDv2 hot paths on CPU, following the reference's algorithm module by module, as a
provider namespace for `bench_workloads.FSDPath / FSDv2Path` (the SAME wiring code then runs on these CPU modules):
bench.py's ``cpu_baseline`` (kind "port") and ``parity`` checker for BASELINE.json configs[3] / configs[4], and the CPU
side of tests/test_fsd_chain.py.  Nothing under sst_amd/ imports this file.

Restated, with the reference's constructor arguments and sub-module names (so a ``state_dict`` of the sst_amd modules, or
of the reference's own, loads with ``strict=True``):

  voxelize            DynamicVoxelNet.voxelize / VoteSegmentor.voxelize     detectors/single_stage_fsd.py:206-226
  scatter_v2          torch.unique(dim=0) + segmented reduce                ops/sst/sst_ops.py:151-182
  DynamicScatterVFE   decorate, [Linear -> BN1d -> ReLU -> scatter max -> gather + concat] x L   voxel_encoders/voxel_encoder.py:502-612,
                                                                            utils.py:107-144
  SubM / strided / inverse sparse convolution = per kernel offset gather -> mm -> scatter-add (the reference's own CPU
                      formulation)                                          ops/spconv/include/spconv/spconv_ops.h:256-357;
                      rulebook = oracle/spconv_oracle.indice_pairs (pinned to the reference's compiled CPU templates)
  SparseBasicBlock, make_sparse_convmodule                                  ops/sparse_block.py:83-141, 218-289
  SimpleSparseUNet, VirtualVoxelMixer                                       middle_encoders/sparse_unet.py:324-504
  PseudoMiddleEncoderForSpconvFSD                                           middle_encoders/sst_input_layer_v2.py:15-37
  ClusterAssigner (dense adjacency -> scipy connected components)           detectors/single_stage_fsd.py:30-84, 144-151, 922-999
  SIRLayer, SIR                                                             voxel_encoders/voxel_encoder.py:617-764, backbones/sir.py:15-88
  VirtualVoxelExtractor = SingleStageFSDV2.extract_feat                     detectors/single_stage_fsd_v2.py:107-129, 159-271
  DynamicPointROIExtractor over oracle/point_pool_oracle (PARITY UNPINNED beyond membership, see that file)

PINNED: tests/test_fsd_chain.py runs every class here against the reference's own Python (oracle/ref_loader) with copied
weights in the build container, and against the committed chain goldens (tests/golden/fsd_chain.npz, fsdv2_chain.npz,
produced by the reference) everywhere."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import point_pool_oracle, spconv_oracle, voxel_oracle


# ----------------------------------------------------------------------------------------------------------- primitives
def voxelize(points_list, voxel_size, point_cloud_range):
    """-> (points [N, C] of all samples, coors [N, 4] int64 (b, z, y, x)); voxelization_cpu.cpp:7-41 via the oracle"""
    coors = [np.pad(voxel_oracle.dynamic_voxelize(p[:, :3].float().contiguous().numpy(), voxel_size, point_cloud_range),
                    ((0, 0), (1, 0)), constant_values=b) for b, p in enumerate(points_list)]
    return torch.cat(points_list), torch.from_numpy(np.concatenate(coors)).long()


def _segment(src, index, n_out, reduce):
    idx = index.view(-1, 1).expand_as(src)
    if reduce == 'max':
        out = torch.full((n_out, src.size(1)), float('-inf'), dtype=src.dtype)
        return out.scatter_reduce(0, idx, src, reduce='amax', include_self=True)
    out = torch.zeros((n_out, src.size(1)), dtype=src.dtype)
    return out.scatter_reduce(0, idx, src, reduce='sum' if reduce == 'sum' else 'mean', include_self=False)


def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None):
    """sst_ops.py:151-182"""
    assert feat.size(0) == coors.size(0)
    mode = 'mean' if mode == 'avg' else mode
    if unq_inv is None:
        new_coors, unq_inv, cnt = torch.unique(coors, return_inverse=True, return_counts=True, dim=0)
    else:
        assert new_coors is not None
    if min_points > 0:
        keep = cnt[unq_inv] >= min_points
        feat, coors = feat[keep], coors[keep]
        new_coors, unq_inv = torch.unique(coors, return_inverse=True, dim=0)
    assert mode in ('max', 'mean', 'sum')
    new_feat = _segment(feat, unq_inv, new_coors.size(0), mode)
    return (new_feat, new_coors, unq_inv) if return_inv else (new_feat, new_coors)


def _norm(cfg, channels):
    cfg = dict(cfg)
    kind = cfg.pop('type')
    cfg.pop('requires_grad', None)
    if kind == 'LN':
        return nn.LayerNorm(channels, eps=cfg.get('eps', 1e-5))
    assert kind in ('BN1d', 'naiveSyncBN1d', 'BN'), kind    # one process: naiveSyncBN1d == BatchNorm1d (ops/norm.py:54-86)
    return nn.BatchNorm1d(channels, eps=cfg.get('eps', 1e-5), momentum=cfg.get('momentum', 0.1))


def _act(name):
    return {'relu': nn.ReLU, 'gelu': nn.GELU}[name]()


def build_mlp(in_channel, hidden_dims, norm_cfg, is_head=False, act='relu', bias=False, dropout=0):
    """sst_ops.py:334-361"""
    layers, last = [], in_channel
    hidden_dims = [hidden_dims] if isinstance(hidden_dims, int) else hidden_dims
    for i, c in enumerate(hidden_dims):
        if i == len(hidden_dims) - 1 and is_head:
            layers.append(nn.Linear(last, c, bias=True))
        else:
            seq = [nn.Linear(last, c, bias=bias), _norm(norm_cfg, c), _act(act)]
            if dropout > 0:
                seq.append(nn.Dropout(dropout))
            layers.append(nn.Sequential(*seq))
        last = c
    return nn.Sequential(*layers)


class _VFELayer(nn.Module):
    """DynamicVFELayer / DynamicVFELayerV2 (utils.py:107-189): Linear(no bias) -> norm -> activation"""

    def __init__(self, cin, cout, norm_cfg, act='relu'):
        super().__init__()
        self.norm = _norm(norm_cfg, cout)
        self.linear = nn.Linear(cin, cout, bias=False)
        self.act_name = act

    def forward(self, x):
        y = self.norm(self.linear(x))
        return F.relu(y) if self.act_name == 'relu' else F.gelu(y)


# ------------------------------------------------------------------------------------------------------ voxel encoders
class DynamicScatterVFE(nn.Module):
    """voxel_encoder.py:502-612"""

    def __init__(self, in_channels=4, feat_channels=(), with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, unique_once=False):
        super().__init__()
        assert not with_distance and fusion_layer is None
        cin = in_channels + 3 * int(with_cluster_center) + 3 * int(with_voxel_center)
        self.with_cluster_center, self.with_voxel_center = with_cluster_center, with_voxel_center
        self.vx, self.vy, self.vz = voxel_size
        r = point_cloud_range
        self.x_offset, self.y_offset, self.z_offset = self.vx / 2 + r[0], self.vy / 2 + r[1], self.vz / 2 + r[2]
        chans = [cin] + list(feat_channels)
        self.vfe_layers = nn.ModuleList(_VFELayer(chans[i] * (2 if i > 0 else 1), chans[i + 1], norm_cfg)
                                        for i in range(len(chans) - 1))
        self.mode, self.rel_dist_scaler, self.return_point_feats = mode, rel_dist_scaler, return_point_feats

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False):
        new_coors, inv = torch.unique(coors, return_inverse=True, dim=0)   # unique_once or not: the same grouping
        parts = [features]
        if self.with_cluster_center:
            mean, _, _ = scatter_v2(features[:, :3], coors, 'avg', new_coors=new_coors, unq_inv=inv)
            parts.append((features[:, :3] - mean[inv]) / self.rel_dist_scaler)
        if self.with_voxel_center:
            parts.append(torch.stack([features[:, 0] - (coors[:, 3].type_as(features) * self.vx + self.x_offset),
                                      features[:, 1] - (coors[:, 2].type_as(features) * self.vy + self.y_offset),
                                      features[:, 2] - (coors[:, 1].type_as(features) * self.vz + self.z_offset)], 1))
        feats = torch.cat(parts, -1)
        for i, layer in enumerate(self.vfe_layers):
            point_feats = layer(feats)
            voxel_feats, voxel_coors, _ = scatter_v2(point_feats, coors, self.mode, new_coors=new_coors, unq_inv=inv)
            if i != len(self.vfe_layers) - 1:
                feats = torch.cat([point_feats, voxel_feats[inv]], 1)
        if self.return_point_feats:
            return point_feats
        return (voxel_feats, voxel_coors, inv) if return_inv else (voxel_feats, voxel_coors)


class SIRLayer(nn.Module):
    """voxel_encoder.py:617-764 (act != 'relu' branch of the constructor: DynamicVFELayerV2 layers)"""

    def __init__(self, in_channels, feat_channels, rel_mlp_hidden_dims, norm_cfg, mode='max', with_cluster_center=False,
                 with_rel_mlp=True, rel_mlp_in_channel=3, return_point_feats=False, rel_dist_scaler=10.0,
                 with_shortcut=True, xyz_normalizer=(1.0, 1.0, 1.0), act='relu'):
        super().__init__()
        self.with_cluster_center, self.with_rel_mlp = with_cluster_center, with_rel_mlp
        cin = in_channels + 3 * int(with_cluster_center)
        if with_rel_mlp:
            self.rel_mlp = build_mlp(rel_mlp_in_channel, list(rel_mlp_hidden_dims) + [in_channels], norm_cfg, act=act)
        chans = [cin] + list(feat_channels)
        self.vfe_layers = nn.ModuleList(_VFELayer(chans[i] * (2 if i > 0 else 1), chans[i + 1], norm_cfg, act)
                                        for i in range(len(chans) - 1))
        self.mode, self.rel_dist_scaler, self.with_shortcut = mode, rel_dist_scaler, with_shortcut
        self.xyz_normalizer, self.return_point_feats = list(xyz_normalizer), return_point_feats

    def forward(self, features, coors, f_cluster=None, return_both=False, unq_inv_once=None, new_coors_once=None):
        xyz_n = torch.tensor(self.xyz_normalizer, dtype=features.dtype)
        head = torch.cat([features[:, :3] / xyz_n[None], features[:, 3:]], 1)
        shortcut = features[:, 3:]
        if f_cluster is None:
            mean, _, inv = scatter_v2(features[:, :3], coors, 'avg', unq_inv=unq_inv_once, new_coors=new_coors_once)
            f_cluster = (features[:, :3] - mean[inv]) / self.rel_dist_scaler
        else:
            f_cluster = f_cluster / self.rel_dist_scaler
        parts = [head]
        if self.with_cluster_center:
            parts.append(f_cluster / 10.0)
        if self.with_rel_mlp:
            parts[0] = parts[0] * self.rel_mlp(f_cluster)
        feats = torch.cat(parts, -1)
        pooled = []
        for i, layer in enumerate(self.vfe_layers):
            point_feats = layer(feats)
            voxel_feats, voxel_coors, inv = scatter_v2(point_feats, coors, self.mode, unq_inv=unq_inv_once,
                                                       new_coors=new_coors_once)
            pooled.append(voxel_feats)
            if i != len(self.vfe_layers) - 1:
                feats = torch.cat([point_feats, voxel_feats[inv]], 1)
        voxel_feats = torch.cat(pooled, 1)
        if self.with_shortcut and point_feats.shape == shortcut.shape:
            point_feats = point_feats + shortcut
        if return_both:
            return point_feats, voxel_feats, voxel_coors
        return point_feats, voxel_feats


class SIR(nn.Module):
    """backbones/sir.py:15-88"""

    def __init__(self, num_blocks=5, in_channels=(), feat_channels=(), rel_mlp_hidden_dims=(), with_rel_mlp=True,
                 with_distance=False, with_cluster_center=False, norm_cfg=dict(type='LN', eps=1e-3), mode='max',
                 xyz_normalizer=(1.0, 1.0, 1.0), act='relu', dropout=0, unique_once=False):
        super().__init__()
        assert not with_distance and dropout == 0
        self.num_blocks = num_blocks
        self.block_list = nn.ModuleList(
            SIRLayer(in_channels[i], feat_channels[i], rel_mlp_hidden_dims[i], norm_cfg, mode, with_cluster_center,
                     with_rel_mlp, return_point_feats=i != num_blocks - 1, rel_dist_scaler=10.0,
                     xyz_normalizer=xyz_normalizer, act=act) for i in range(num_blocks))

    def forward(self, points, features, coors, f_cluster=None):
        new_coors, inv = torch.unique(coors, return_inverse=True, dim=0)
        out, pooled, out_coors = features, [], None
        for i, block in enumerate(self.block_list):
            x = torch.cat([points, out], 1)
            if i < self.num_blocks - 1:
                out, cluster = block(x, coors, f_cluster, unq_inv_once=inv, new_coors_once=new_coors)
            else:
                out, cluster, out_coors = block(x, coors, f_cluster, return_both=True, unq_inv_once=inv,
                                                new_coors_once=new_coors)
            pooled.append(cluster)
        return out, torch.cat(pooled, 1), out_coors


# --------------------------------------------------------------------------------------------------- sparse convolution
class SparseTensor(object):
    """features [N, C], indices [N, 4] int32 (b, z, y, x), spatial shape, batch size, rulebooks by indice_key
    (ops/spconv/structure.py:21-73)"""

    def __init__(self, features, indices, spatial_shape, batch_size, rulebooks=None):
        self.features, self.indices = features, indices
        self.spatial_shape, self.batch_size = list(spatial_shape), batch_size
        self.rulebooks = {} if rulebooks is None else rulebooks

    def replace_feature(self, features):
        return SparseTensor(features, self.indices, self.spatial_shape, self.batch_size, self.rulebooks)


def _t3(v):
    return [int(e) for e in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


class SparseConv(nn.Module):
    """SubMConv3d / SparseConv3d / SparseInverseConv3d (ops/spconv/conv.py:49-230) with the arithmetic of
    indiceConv's CPU path (spconv_ops.h:305-350): out[pairs[k][1]] += in[pairs[k][0]] @ W[k]"""

    def __init__(self, kind, in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0):
        super().__init__()
        self.kind, self.indice_key = kind, indice_key
        self.ksize, self.stride, self.padding = _t3(kernel_size), _t3(stride), _t3(padding)
        self.weight = nn.Parameter(torch.empty(*self.ksize, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x):
        rb = x.rulebooks.get(self.indice_key)
        if self.kind == 'inverse':
            outids, in_shape, pairs, num = rb['in_ids'], rb['in_shape'], rb['pairs'], rb['num']
            out_shape, src_row, dst_row = in_shape, 1, 0
        else:
            if rb is None:
                outids, pairs, num, out_shape = spconv_oracle.indice_pairs(
                    x.indices.numpy(), x.batch_size, x.spatial_shape, self.ksize, self.stride, self.padding, [1, 1, 1],
                    subm=self.kind == 'subm')
                rb = x.rulebooks[self.indice_key] = dict(
                    outids=torch.from_numpy(outids), pairs=torch.from_numpy(pairs).long(), num=num.tolist(),
                    out_shape=out_shape, in_ids=x.indices, in_shape=x.spatial_shape)
            outids, pairs, num, out_shape = rb['outids'], rb['pairs'], rb['num'], rb['out_shape']
            src_row, dst_row = 0, 1
        w = self.weight.reshape(-1, self.weight.shape[-2], self.weight.shape[-1])
        out = x.features.new_zeros((outids.size(0), w.size(2)))
        for k in range(w.size(0)):
            c = num[k]
            if c:
                out = out.index_add(0, pairs[k, dst_row, :c], x.features[pairs[k, src_row, :c]] @ w[k])
        return SparseTensor(out, outids, out_shape, x.batch_size, x.rulebooks)


class ConvModule(nn.Sequential):
    """make_sparse_convmodule (ops/sparse_block.py:218-289), order ('conv', 'norm', 'act'): children 0, 1, 2"""

    def __init__(self, cin, cout, kernel_size, indice_key, norm_cfg, kind='subm', stride=1, padding=0):
        super().__init__(SparseConv(kind, cin, cout, kernel_size, indice_key, stride, padding), _norm(norm_cfg, cout),
                         nn.ReLU())

    def forward(self, x):
        y = self[0](x)
        return y.replace_feature(self[2](self[1](y.features)))


class SparseBasicBlock(nn.Module):
    """ops/sparse_block.py:83-141"""

    def __init__(self, channels, indice_key, norm_cfg):
        super().__init__()
        self.conv1 = SparseConv('subm', channels, channels, 3, indice_key, padding=1)
        self.bn1 = _norm(norm_cfg, channels)
        self.conv2 = SparseConv('subm', channels, channels, 3, indice_key, padding=1)
        self.bn2 = _norm(norm_cfg, channels)

    def forward(self, x):
        y = self.conv1(x)
        y = self.conv2(y.replace_feature(F.relu(self.bn1(y.features))))
        return y.replace_feature(F.relu(self.bn2(y.features) + x.features))


class _UNet(nn.Module):
    """the stages SimpleSparseUNet and VirtualVoxelMixer share (middle_encoders/sparse_unet.py:324-504)"""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128, ndim=3,
                 encoder_channels=None, encoder_paddings=None, decoder_channels=None, decoder_paddings=None,
                 keep_coors_dims=None, act_type='relu', **unused):
        super().__init__()
        assert tuple(order) == ('conv', 'norm', 'act') and ndim == 3 and act_type == 'relu'
        self.sparse_shape, self.keep_coors_dims = list(sparse_shape), keep_coors_dims
        self.stage_num = len(encoder_channels)
        self.conv_input = ConvModule(in_channels, base_channels, 3, 'subm1', norm_cfg, padding=1)
        self.encoder_layers = nn.Module()
        width = base_channels
        for level, (chans, pads) in enumerate(zip(encoder_channels, encoder_paddings), start=1):
            stage = []
            for j, (out, pad) in enumerate(zip(tuple(chans), tuple(pads))):
                down = level > 1 and j == 0
                stage.append(ConvModule(width, out, 3, f'spconv{level}' if down else f'subm{level}', norm_cfg,
                                        kind='conv' if down else 'subm', stride=2 if down else 1, padding=pad))
                width = out
            self.encoder_layers.add_module(f'encoder_layer{level}', nn.Sequential(*stage))
        for level, ((c_lat, c_merge, c_up), pads) in zip(range(len(decoder_channels), 0, -1),
                                                         zip(decoder_channels, decoder_paddings)):
            assert c_lat == width
            setattr(self, f'lateral_layer{level}', SparseBasicBlock(width, f'subm{level}', norm_cfg))
            setattr(self, f'merge_layer{level}', ConvModule(2 * width, c_merge, 3, f'subm{level}', norm_cfg, padding=pads[0]))
            setattr(self, f'upsample_layer{level}',
                    ConvModule(width, c_up, 3, f'spconv{level}', norm_cfg, kind='inverse') if level > 1
                    else ConvModule(width, c_up, 3, 'subm1', norm_cfg, padding=pads[1]))
            width = c_up
        self.out_width = width

    def _run(self, feats, coors, batch_size):
        x = self.conv_input(SparseTensor(feats, coors.int(), self.sparse_shape, batch_size))
        levels = []
        for level in range(1, self.stage_num + 1):
            for module in getattr(self.encoder_layers, f'encoder_layer{level}'):
                x = module(x)
            levels.append(x)
        self.decoder_features = []                        # every decoder level's output, coarsest first (:367-370)
        for level in range(self.stage_num, 0, -1):      # decoder_layer_forward, sparse_unet.py:161-202
            lat = getattr(self, f'lateral_layer{level}')(levels[level - 1])
            cat = lat.replace_feature(torch.cat([x.features, lat.features], 1))
            merged = getattr(self, f'merge_layer{level}')(cat)
            n, c_out = merged.features.shape
            folded = cat.features.view(n, c_out, -1).sum(2)
            x = getattr(self, f'upsample_layer{level}')(cat.replace_feature(merged.features + folded))
            self.decoder_features.append(x)
        return x


class SimpleSparseUNet(_UNet):

    def __init__(self, *args, return_multiscale_features=False, **kw):
        super().__init__(*args, **kw)
        self.return_multiscale_features = return_multiscale_features

    def forward(self, voxel_info):
        coors = voxel_info['voxel_coors']
        if self.keep_coors_dims is not None:
            coors = coors[:, self.keep_coors_dims]
        batch_size = voxel_info.get('batch_size') or int(coors[:, 0].max()) + 1
        x = self._run(voxel_info['voxel_feats'], coors, batch_size)
        return [{'voxel_feats': x.features, 'voxel_coors': x.indices, 'sparse_shape': x.spatial_shape,
                 'batch_size': x.batch_size,
                 'decoder_features': list(self.decoder_features) if self.return_multiscale_features else []}]


class VirtualVoxelMixer(_UNet):

    def __init__(self, in_channels, sparse_shape, output_channels=128, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 **kw):
        super().__init__(in_channels, sparse_shape, norm_cfg=norm_cfg, output_channels=output_channels, **kw)
        self.conv_out = ConvModule(self.out_width, output_channels, 3, 'out_conv', norm_cfg, padding=0)

    def forward(self, voxel_features, coors, batch_size):
        if self.keep_coors_dims is not None:
            coors = coors[:, self.keep_coors_dims]
        x = self.conv_out(self._run(voxel_features, coors, batch_size))
        return x.features, x.indices, x.spatial_shape


class PseudoMiddleEncoderForSpconvFSD(nn.Module):
    """sst_input_layer_v2.py:15-37"""

    def forward(self, voxel_feats, voxel_coors, batch_size=None):
        info = {'voxel_feats': voxel_feats, 'voxel_coors': voxel_coors.int()}
        if batch_size is not None:
            info['batch_size'] = batch_size
        return info


# ------------------------------------------------------------------------------------------------------------ clustering
class ClusterAssigner(nn.Module):
    """single_stage_fsd.py:922-999 with filter_almost_empty :30-34, find_connected_componets(_single_batch) :45-84 (dense
    N x N adjacency, scipy), modify_cluster_by_class :144-151"""

    def __init__(self, cluster_voxel_size, min_points, point_cloud_range, connected_dist,
                 class_names=('Car', 'Cyclist', 'Pedestrian'), gpu_clustering=(False, False)):
        super().__init__()
        self.cluster_voxel_size, self.min_points, self.connected_dist = cluster_voxel_size, min_points, connected_dist
        self.point_cloud_range, self.class_names = point_cloud_range, list(class_names)

    def _pick(self, table, name):
        if isinstance(table, dict):
            return table[name]
        return table[self.class_names.index(name)] if isinstance(table, list) else table

    @staticmethod
    def _components(centers, batch, dist, per_sample):
        from scipy.sparse.csgraph import connected_components
        out = torch.zeros_like(batch) - 1
        base = 0
        groups = [batch == b for b in range(int(batch.max()) + 1)] if per_sample else [torch.ones_like(batch, dtype=torch.bool)]
        for msk in groups:
            if not bool(msk.any()):
                continue
            p = centers[msk]
            d = ((p[:, None, :2] - p[None, :, :2]) ** 2).sum(2) ** 0.5
            lab = torch.from_numpy(connected_components((d < dist).numpy(), directed=False)[1]).int() + base
            base = int(lab.max()) + 1 if per_sample else base
            out[msk] = lab
        return out

    @torch.no_grad()
    def forward(self, points_list, batch_idx_list, gt_bboxes_3d=None, gt_labels_3d=None, origin_points=None):
        inds, valids = [], []
        for c, (points, batch_idx, name) in enumerate(zip(points_list, batch_idx_list, self.class_names)):
            batch_idx = batch_idx.int()
            points = points.float()               # integer stage: fp32 arithmetic whatever the dtype of the evaluation
            vs = torch.tensor(self._pick(self.cluster_voxel_size, name))
            lo = torch.tensor(self.point_cloud_range[:3], dtype=points.dtype)
            coors = torch.div(points - lo[None], vs[None], rounding_mode='floor').int()
            coors = torch.cat([batch_idx[:, None], coors], 1)
            _, inv, cnt = torch.unique(coors, return_inverse=True, return_counts=True, dim=0)
            valid = cnt[inv] >= self.min_points
            if not bool(valid.any()):
                valid = ~valid
            points, batch_idx, coors = points[valid], batch_idx[valid], coors[valid]
            centers, vcoors, inv = scatter_v2(points, coors, 'avg')
            comp = self._components(centers, vcoors[:, 0], self._pick(self.connected_dist, name), self.training)
            per_point = torch.stack([batch_idx, comp[inv]], 1)
            inds.append(torch.cat([per_point.new_full((len(per_point), 1), c), per_point], 1))
            valids.append(valid)
        return inds, valids


# ----------------------------------------------------------------------------------------------------- RoI point pooling
class DynamicPointROIExtractor(nn.Module):
    """roi_extractors/dynamic_point_roi_extractor.py:9-136 over the numpy restatement of the pool (TorchEx source absent:
    parity unpinned beyond membership, oracle/point_pool_oracle.py)"""

    def __init__(self, extra_wlh=(0, 0, 0), max_inbox_point=512, max_all_pts=50000, debug=True, init_cfg=None):
        super().__init__()
        self.extra_wlh, self.max_inbox_point, self.max_all_pts = list(extra_wlh), max_inbox_point, max_all_pts

    def forward(self, pts_xyz, batch_inds, rois, max_inbox_point=None, batch_size=None):
        cap = self.max_inbox_point if max_inbox_point is None else max_inbox_point
        n_samples = int(batch_size) if batch_size is not None else int(batch_inds.max()) + 1
        pieces = []
        for b in range(n_samples):
            p_sel = torch.nonzero(batch_inds == b).squeeze(1)
            r_sel = torch.nonzero(rois[:, 0].long() == b).squeeze(1)
            if len(p_sel) and len(r_sel):
                # fp32 whatever the dtype of the evaluation: membership is a decision, the oracle's arithmetic is fp32
                pi, ri, ft = point_pool_oracle.dynamic_point_pool(rois[r_sel, 1:].float().numpy(), pts_xyz[p_sel].float().numpy(),
                                                                  self.extra_wlh, cap, self.max_all_pts)
                if len(pi) and pi[0] >= 0:
                    pieces.append((p_sel[torch.from_numpy(np.asarray(pi)).long()], r_sel[torch.from_numpy(np.asarray(ri)).long()],
                                   torch.from_numpy(np.asarray(ft, dtype=np.float32))))
                    continue
            pieces.append((torch.full((1,), -1, dtype=torch.long), torch.full((1,), -1, dtype=torch.long),
                           torch.zeros((1, 13))))
        inds, roi_inds, info = (torch.cat(col) for col in zip(*pieces))
        info = info.to(pts_xyz.dtype)
        return inds, roi_inds, dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1], is_in_margin=info[:, -1])


# -------------------------------------------------------------------------------------------------- FSDv2 virtual voxels
class VirtualVoxelExtractor(nn.Module):
    """SingleStageFSDV2.extract_feat, non-baseline mode (single_stage_fsd_v2.py:159-271) with voxelize_with_batch_idx
    :107-121 and clip_points :124-129; constructor = the detector's sub-configs"""

    def __init__(self, backbone, voxel_encoder, virtual_point_projector, train_cfg=None, test_cfg=None, multiscale_cfg=None,
                 bbox_head=None, as_rpn=None):
        super().__init__()
        ve = dict(voxel_encoder)
        assert ve.pop('type') == 'DynamicScatterVFE'
        self.voxel_encoder = DynamicScatterVFE(**ve)
        self.virtual_voxel_size, self.point_cloud_range = ve['voxel_size'], ve['point_cloud_range']
        bb = dict(backbone)
        assert bb.pop('type') == 'VirtualVoxelMixer'
        self.backbone = VirtualVoxelMixer(**bb)
        vpp = virtual_point_projector
        self.virtual_proj = build_mlp(vpp['in_channels'], vpp['hidden_dims'], vpp['norm_cfg'])
        self.ori_proj = build_mlp(vpp['ori_in_channels'], vpp['ori_hidden_dims'], vpp['norm_cfg'])
        self.zero_virtual_feature = vpp.get('zero_virtual_feature', False)
        self.only_virtual = vpp.get('only_virtual', False)
        self.as_rpn = bool((bbox_head or {}).get('as_rpn', False)) if as_rpn is None else bool(as_rpn)
        if self.as_rpn:                                                                    # :92-93
            self.recover_proj = build_mlp(vpp['recover_in_channels'], vpp['recover_hidden_dims'], vpp['norm_cfg'])
        self.multiscale_cfg = multiscale_cfg
        if multiscale_cfg is not None:                                                     # :99-105
            self.ms_projectors = nn.ModuleList([build_mlp(p[0], p[1:], multiscale_cfg['norm_cfg'])
                                                for p in multiscale_cfg['projector_hiddens']])

    def ms_coors_proj(self, coors, sparse_shape):
        """single_stage_fsd_v2.py:399-433"""
        tgt = self.multiscale_cfg['target_sparse_shape']
        bev_stride, z_stride = tgt[1] // sparse_shape[1], tgt[0] // sparse_shape[0]
        assert bev_stride == tgt[2] / sparse_shape[2] and z_stride >= 1 and bev_stride >= 1
        out = coors.clone()
        out[:, 1] = coors[:, 1] * z_stride + z_stride // 2
        out[:, 2] = coors[:, 2] * bev_stride + bev_stride // 2
        out[:, 3] = coors[:, 3] * bev_stride + bev_stride // 2
        assert int(out[:, 1].max()) < tgt[0] and int(out[:, 2].max()) < tgt[1] and int(out[:, 3].max()) < tgt[2]
        return out

    def multiscale_fusion(self, ms_data, voxel_feats, coors):
        """single_stage_fsd_v2.py:375-397: projected decoder features of the segmentor join the virtual voxels; two
        groupings of the concatenated coordinates (feature average, indicator maximum), as the reference does"""
        cfg = self.multiscale_cfg
        ms_data = [ms_data[lvl] for lvl in cfg['multiscale_levels']]
        ms_feats = [self.ms_projectors[i](d.features) for i, d in enumerate(ms_data)]
        ms_coors = [self.ms_coors_proj(d.indices, d.spatial_shape) for d in ms_data]
        n_add = sum(len(f) for f in ms_feats)
        cat_feats = torch.cat([voxel_feats] + ms_feats, 0)
        cat_coors = torch.cat([coors] + ms_coors, 0)
        indicators = torch.cat([voxel_feats.new_ones(len(voxel_feats), 1), voxel_feats.new_zeros(n_add, 1)], 0)
        out_feats, out_coors = scatter_v2(cat_feats, cat_coors, cfg['fusion_mode'], return_inv=False)
        out_ind, _ = scatter_v2(indicators, cat_coors, 'max', return_inv=False)
        mask = out_ind.squeeze(1) == 1
        assert int(mask.sum()) == len(voxel_feats)
        return out_feats, out_coors, mask

    def voxelize_with_batch_idx(self, points, batch_idx):
        xyz = points[:, :3].float()                       # @force_fp32 in the reference; a float64 evaluation of the port
        vs = xyz.new_tensor(self.virtual_voxel_size)      # (the gradient adjudication) keeps the integer stage of the f32 run
        lo = xyz.new_tensor(self.point_cloud_range[:3])
        cells = torch.div(xyz - lo[None], vs[None], rounding_mode='floor').long()
        return torch.cat([batch_idx[:, None], cells[:, [2, 1, 0]]], 1)

    def forward(self, sampled_dict, origin_dict, gt_bboxes_3d=None, multiscale_features=None):
        fg_pts, fg_batch = sampled_dict['seg_points'], sampled_dict['batch_idx']
        r = self.point_cloud_range
        centers = sampled_dict['center_preds']
        for d in range(3):      # clip_points: in place on the caller's tensor
            centers[:, d] = centers[:, d].clamp(min=r[d] + 1e-5, max=r[d + 3] - 1e-5)
        offset = (centers - fg_pts[:, :3]) / 10
        vir_feat = self.virtual_proj(torch.cat([sampled_dict['seg_feats'], offset, sampled_dict['seg_logits'],
                                                fg_pts[:, 3:]], 1))
        if self.zero_virtual_feature:
            vir_feat = vir_feat * 0
        ori_pts = origin_dict['seg_points']
        ori_feat = self.ori_proj(origin_dict['seg_feats'])
        cat_pts = torch.cat([ori_pts[:, :3], centers], 0)
        cat_feat = torch.cat([ori_feat, vir_feat], 0)
        cat_batch = torch.cat([origin_dict['batch_idx'], fg_batch], 0)
        coors = self.voxelize_with_batch_idx(cat_pts, cat_batch)
        voxel_feats, voxel_coors, unq_inv = self.voxel_encoder(torch.cat([cat_pts, cat_feat], 1), coors, return_inv=True)
        encoder_coors = voxel_coors
        indicator = torch.cat([cat_pts.new_zeros(ori_pts.size(0)), cat_pts.new_ones(centers.size(0))])
        share, scoors = scatter_v2(indicator[:, None], coors, 'avg', return_inv=False)
        assert bool((scoors == voxel_coors).all())
        virtual = share[:, 0] > 0
        batch_size = int(voxel_coors[:, 0].max()) + 1
        single = None
        if multiscale_features is not None:                                                # :208-209
            voxel_feats, voxel_coors, single = self.multiscale_fusion(multiscale_features, voxel_feats, voxel_coors)
        if self.only_virtual:
            assert multiscale_features is None
            voxel_feats, voxel_coors = voxel_feats[virtual], voxel_coors[virtual]
        out_feats, out_coors, sparse_shape = self.backbone(voxel_feats, voxel_coors, batch_size)
        if single is not None:                                                             # :218-221
            out_feats, out_coors = out_feats[single], out_coors[single]
        vs = cat_pts.new_tensor(self.virtual_voxel_size)
        lo = cat_pts.new_tensor(self.point_cloud_range[:3])
        voxel_centers = (out_coors[:, [3, 2, 1]] + 0.5) * vs[None] + lo[None]
        pick = slice(None) if self.only_virtual else virtual
        out = dict(virtual_feats=out_feats[pick], virtual_coors=out_coors[pick], virtual_centers=voxel_centers[pick],
                   sparse_shape=sparse_shape)
        if self.training:
            centroid, _ = scatter_v2(cat_pts[:, :3], coors, 'avg', return_inv=False)
            out['virtual_centroid'] = centroid[virtual]
        if self.as_rpn:                                                                    # :131-155, 263-270
            assert bool((out_coors == encoder_coors).all())
            center_per_pts = (out_coors[unq_inv][:, [3, 2, 1]] + 0.5) * vs[None] + lo[None]
            offset = (center_per_pts - cat_pts) / vs[None] * 2
            out['pts_feats'] = self.recover_proj(torch.cat([out_feats[unq_inv], offset], 1))
            out['pts_xyz'], out['pts_indicators'], out['pts_batch_inds'] = cat_pts, indicator, cat_batch
        return out
