"""Generates the golden fixtures in this directory by EXECUTING THE REFERENCE (build container only):

  * the reference's own C++ dynamic_voxelize (oracle/_ref/voxel_layer_ref.so, compiled from
    /root/reference/mmdet3d/ops/voxel/src by oracle/build_ref.py);
  * the reference's own Python for SSTInputLayerV2 / SSTv2 blocks / DynamicVFE / SIR, loaded unmodified by
    file path under stubs (oracle/ref_loader.py: TorchEx ingroup_indices -> stable rank, torch_scatter ->
    scatter_reduce, DynamicScatter -> oracle restatement of scatter_points_cuda.cu).

Run:  python tests/golden/make_golden.py          (from the repo root; needs /root/reference)
The fixtures are small (< 1.5 MB each) .npz files committed next to this script; the GPU box only reads them.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import build_ref, ref_loader  # noqa: E402

VOXEL_SIZE = (0.32, 0.32, 6)
PC_RANGE = [-74.88, -74.88, -2, 74.88, 74.88, 4]
DROP_TRAIN = {
    0: {'max_tokens': 30, 'drop_range': (0, 30)},
    1: {'max_tokens': 60, 'drop_range': (30, 60)},
    2: {'max_tokens': 100, 'drop_range': (60, 100000)},
}
DROP_TEST = {
    0: {'max_tokens': 30, 'drop_range': (0, 30)},
    1: {'max_tokens': 60, 'drop_range': (30, 60)},
    2: {'max_tokens': 100, 'drop_range': (60, 100)},
    3: {'max_tokens': 144, 'drop_range': (100, 100000)},
}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def t2n(t):
    return t.detach().cpu().numpy()


def state_to_np(sd, prefix='w::'):
    return {prefix + k: t2n(v) for k, v in sd.items()}


def gen_voxelize():
    mod = build_ref.load()
    assert mod is not None, 'build oracle/_ref first (python oracle/build_ref.py)'
    g = torch.Generator().manual_seed(0)
    cases = {}
    # SST Waymo grid, with 8 % of the points outside the range (exercises the clamp) and points ON the borders
    pts = torch.rand(4000, 5, generator=g) * torch.tensor([170.0, 170.0, 8.0, 1, 1]) + torch.tensor(
        [-85.0, -85.0, -3.0, 0, 0])
    pts[:8, :3] = torch.tensor([[-74.88, -74.88, -2.0], [74.88, 74.88, 4.0], [74.8799, 0, 0], [-74.8801, 0, 0],
                                [0, 74.56, 3.999], [0.32, 0.32, 0], [-0.32, -0.32, 0], [1e6, -1e6, 50.0]])
    cases['sst'] = (pts, list(VOXEL_SIZE), PC_RANGE)
    # FSD segmentor grid (3-D voxels)
    pts2 = torch.rand(3000, 4, generator=g) * torch.tensor([160.0, 160.0, 7.0, 1]) + torch.tensor(
        [-80.0, -80.0, -2.5, 0])
    cases['fsd'] = (pts2, [0.25, 0.25, 0.2], [-80, -80, -2, 80, 80, 4])
    # FSDv2 nuScenes grid
    pts3 = torch.rand(3000, 3, generator=g) * torch.tensor([110.0, 110.0, 9.0]) + torch.tensor([-55.0, -55.0, -5.5])
    cases['fsdv2'] = (pts3, [0.2, 0.2, 0.2], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
    out = {}
    for name, (p, vs, rng) in cases.items():
        coors = torch.zeros((p.size(0), 3), dtype=torch.int32)
        mod.dynamic_voxelize(p.contiguous(), coors, vs, rng, 3)
        out[f'{name}::points'] = t2n(p)
        out[f'{name}::voxel_size'] = np.asarray(vs, dtype=np.float64)
        out[f'{name}::range'] = np.asarray(rng, dtype=np.float64)
        out[f'{name}::coors'] = t2n(coors)
    save('voxelize.npz', **out)


def make_voxel_coors(g, n_pts, batch, crowded=False):
    """unique (b,z,y,x) voxel coordinates, sorted lexicographically like the voxel encoder emits them."""
    rows = []
    for b in range(batch):
        if crowded:  # concentrate points so that some 12x12 windows hold > 100 voxels
            xy = (torch.randn(n_pts, 2, generator=g) * 14 + 234).clamp(0, 467).long()
        else:
            xy = torch.randint(0, 468, (n_pts, 2), generator=g)
        c = torch.cat([torch.full((n_pts, 1), b), torch.zeros(n_pts, 1, dtype=torch.long), xy[:, 1:2], xy[:, 0:1]], 1)
        rows.append(torch.unique(c, dim=0))
    return torch.cat(rows, 0)


def gen_input_layer(ref):
    cls = ref.input_layer_v2.SSTInputLayerV2
    g = torch.Generator().manual_seed(1)
    for tag, training, crowded, n_pts in (('eval', False, False, 2500), ('train', True, True, 2500)):
        coors = make_voxel_coors(g, n_pts, 2, crowded)
        feats = torch.randn(coors.size(0), 128, generator=g)
        layer = cls(drop_info=(DROP_TRAIN, DROP_TEST), window_shape=(12, 12, 1), sparse_shape=(468, 468, 1),
                    shuffle_voxels=False, debug=True, pos_temperature=10000, normalize_pos=False, mute=True)
        layer.train(training)
        info = layer(feats, coors.int(), 2)
        out = {'in::voxel_coors': t2n(coors).astype(np.int32), 'in::training': np.asarray(int(training))}
        for k in ('voxel_coors', 'voxel_keep_inds'):
            out['out::' + k] = t2n(info[k])
        for s in range(2):
            for k in (f'batch_win_inds_shift{s}', f'coors_in_win_shift{s}', f'voxel_drop_level_shift{s}'):
                out['out::' + k] = t2n(info[k])
            inds = info[f'flat2win_inds_shift{s}']
            m = info['voxel_coors'].size(0)
            f2w = -np.ones(m, dtype=np.int64)
            for dl in inds:
                if isinstance(dl, str):
                    continue
                f2w[t2n(inds[dl][1][0])] = t2n(inds[dl][0])
            out[f'out::flat2win_shift{s}'] = f2w
            pos_flat = ref.sst_ops.window2flat_v2(info[f'pos_dict_shift{s}'], inds)
            out[f'out::pos_flat_shift{s}'] = t2n(pos_flat).astype(np.float32)
            n_pad = sum(int(v.numel()) for v in info[f'key_mask_shift{s}'].values())
            n_true = sum(int(v.sum()) for v in info[f'key_mask_shift{s}'].values())
            out[f'out::key_mask_stats_shift{s}'] = np.asarray([n_pad, n_true])
        print(tag, 'voxels in/out', coors.size(0), info['voxel_coors'].size(0))
        save(f'input_layer_{tag}.npz', **out)


def gen_sst_block(ref, only=None):
    """One BasicShiftBlockV2 (2 encoder layers) through the reference SSTv2, fp32, with gradients.
    'std' is the real SST-base geometry (d=128, 8 heads, FFN 256); the variants use d=64 / 4 heads to keep
    the fixtures small."""
    layer = ref.input_layer_v2.SSTInputLayerV2(drop_info=(DROP_TRAIN, DROP_TEST), window_shape=(12, 12, 1),
                                               sparse_shape=(468, 468, 1), shuffle_voxels=False, debug=True,
                                               mute=True)
    layer.eval()
    variants = (('std', 128, 8, 256, dict()), ('cosine', 64, 4, 128, dict(cosine=True, tau_min=0.01)),
                ('cosine_ns', 64, 4, 128, dict(cosine=True, tau_min=0.01, non_shared_tau=True)),
                ('prenorm', 64, 4, 128, dict(post_norm=False)),
                # FSD's SST encoder (configs/fsd/fsd_waymoD1_1x_sst_encoder.py): batch norm instead of LayerNorm
                ('bn_cosine', 64, 4, 128, dict(use_bn=True, cosine=True, tau_min=0.01)))
    for tag, d, h, ffn, layer_cfg in variants:
        if only is not None and tag not in only:
            continue
        g = torch.Generator().manual_seed(2)
        coors = make_voxel_coors(g, 170, 2, crowded=True)
        m = coors.size(0)
        torch.manual_seed(3)
        net = ref.sst_v2.SSTv2(d_model=[d], nhead=[h], num_blocks=1, dim_feedforward=[ffn], output_shape=[468, 468],
                               num_attached_conv=0, to_bev=False, debug=True, layer_cfg=layer_cfg)
        # biases default to zero in nn.MultiheadAttention: randomise so that the test sees them
        with torch.no_grad():
            for n_, p_ in net.named_parameters():
                if p_.dim() == 1:
                    p_.add_(torch.randn(p_.shape, generator=g) * 0.1)
                if n_.endswith('tau'):
                    p_.copy_(0.5 + torch.rand(p_.shape, generator=g))
        net.train()
        feats = torch.randn(m, d, generator=g).requires_grad_(True)
        info = layer(feats, coors.int(), 2)
        out_feats = net(info)[0]['voxel_feats']
        gout = torch.randn(out_feats.shape, generator=g)
        (out_feats * gout).sum().backward()
        arrays = {'in::voxel_coors': t2n(coors).astype(np.int32), 'in::voxel_feats': t2n(feats),
                  'in::grad_out': t2n(gout), 'out::voxel_feats': t2n(out_feats), 'out::grad_in': t2n(feats.grad),
                  'cfg::d_model': np.asarray(d), 'cfg::nhead': np.asarray(h), 'cfg::ffn': np.asarray(ffn)}
        arrays.update(state_to_np(net.state_dict()))
        for n_, p_ in net.named_parameters():
            if 'encoder_list.0.win_attn.self_attn.in_proj' in n_ or 'encoder_list.1.linear1.weight' in n_ \
                    or n_.endswith('tau'):
                arrays['grad::' + n_] = t2n(p_.grad)
        print(tag, 'voxels', m)
        save(f'sst_block_{tag}.npz', **arrays)


def gen_sst_block_bf16(ref):
    """The 'std' block (d = 128, 8 heads, FFN 256; one BasicShiftBlockV2 = 2 encoder layers) through the reference
    SSTv2 under torch.autocast(bfloat16) on CPU: linear layers and matrix products in bf16, softmax / LayerNorm /
    residual stream in fp32 (torch's autocast policy) - the pin of the reduced-precision mode (sst_amd/bf16.py)."""
    layer = ref.input_layer_v2.SSTInputLayerV2(drop_info=(DROP_TRAIN, DROP_TEST), window_shape=(12, 12, 1),
                                               sparse_shape=(468, 468, 1), shuffle_voxels=False, debug=True,
                                               mute=True)
    layer.eval()
    g = torch.Generator().manual_seed(2)
    coors = make_voxel_coors(g, 170, 2, crowded=True)
    m = coors.size(0)
    torch.manual_seed(3)
    net = ref.sst_v2.SSTv2(d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256], output_shape=[468, 468],
                           num_attached_conv=0, to_bev=False, debug=True)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn(p_.shape, generator=g) * 0.1)
    net.train()
    feats = torch.randn(m, 128, generator=g).requires_grad_(True)
    info = layer(feats, coors.int(), 2)
    with torch.autocast(device_type='cpu', dtype=torch.bfloat16):
        out_feats = net(info)[0]['voxel_feats']
    out_feats = out_feats.float()
    gout = torch.randn(out_feats.shape, generator=g)
    (out_feats * gout).sum().backward()
    # the same network in fp32, for the size of the precision effect itself
    feats32 = feats.detach().clone().requires_grad_(True)
    out32 = net(layer(feats32, coors.int(), 2))[0]['voxel_feats']
    arrays = {'in::voxel_coors': t2n(coors).astype(np.int32), 'in::voxel_feats': t2n(feats), 'in::grad_out': t2n(gout),
              'out::voxel_feats': t2n(out_feats), 'out::grad_in': t2n(feats.grad), 'out::voxel_feats_fp32': t2n(out32)}
    arrays.update(state_to_np(net.state_dict()))
    for n_, p_ in net.named_parameters():
        if 'encoder_list.0.win_attn.self_attn.in_proj' in n_ or 'encoder_list.1.linear1.weight' in n_ \
                or 'encoder_list.1.norm2' in n_:
            arrays['grad::' + n_] = t2n(p_.grad)
    print('bf16 block: voxels', m, 'autocast vs fp32 max abs', float((out_feats - out32).abs().max()))
    save('sst_block_bf16.npz', **arrays)


def gen_sst_bev(ref):
    """The output side of SSTv2 (a14): recover_bev + attached dilated convolutions + BN2d + ReLU (sst_v2.py:86-92,
    139-197), with a 48 x 48 canvas so that the dense output stays a small fixture; also with conv_shortcut."""
    layer = ref.input_layer_v2.SSTInputLayerV2(drop_info=(DROP_TRAIN, DROP_TEST), window_shape=(12, 12, 1),
                                               sparse_shape=(48, 48, 1), shuffle_voxels=False, debug=True, mute=True)
    layer.eval()
    for tag, shortcut in (('plain', False), ('shortcut', True)):
        g = torch.Generator().manual_seed(4)
        rows = []
        for b in range(2):
            xy = torch.randint(0, 48, (500, 2), generator=g)
            c = torch.cat([torch.full((500, 1), b), torch.zeros(500, 1, dtype=torch.long), xy[:, 1:2], xy[:, 0:1]], 1)
            rows.append(torch.unique(c, dim=0))
        coors = torch.cat(rows, 0)
        m = coors.size(0)
        torch.manual_seed(5)
        net = ref.sst_v2.SSTv2(d_model=[32], nhead=[2], num_blocks=1, dim_feedforward=[64], output_shape=[48, 48],
                               num_attached_conv=2, conv_in_channel=32, conv_out_channel=32, debug=True, to_bev=True,
                               conv_shortcut=shortcut)
        net.train()
        feats = torch.randn(m, 32, generator=g).requires_grad_(True)
        info = layer(feats, coors.int(), 2)
        bev = net(info)[0]
        gout = torch.randn(bev.shape, generator=g)
        (bev * gout).sum().backward()
        arrays = {'in::voxel_coors': t2n(coors).astype(np.int32), 'in::voxel_feats': t2n(feats),
                  'in::grad_out': t2n(gout), 'out::bev': t2n(bev), 'out::grad_in': t2n(feats.grad)}
        arrays.update(state_to_np(net.state_dict()))   # after the step: BN running statistics updated once
        for n_, p_ in net.named_parameters():
            if n_.startswith('conv_layer'):
                arrays['grad::' + n_] = t2n(p_.grad)
        print(tag, 'voxels', m, 'bev', tuple(bev.shape))
        save(f'sst_bev_{tag}.npz', **arrays)


def gen_sst_v1(ref):
    """Reference SSTInputLayer (v1) + SSTv1 (2 blocks of d=64 / 4 heads, no attached conv), eval mode, fp32."""
    g = torch.Generator().manual_seed(8)
    coors = make_voxel_coors(g, 200, 2, crowded=True)
    m = coors.size(0)
    layer = ref.input_layer_v1.SSTInputLayer(drop_info=(DROP_TRAIN, DROP_TEST), shifts_list=[(0, 0), (6, 6)],
                                             window_shape=(12, 12), point_cloud_range=PC_RANGE, voxel_size=VOXEL_SIZE,
                                             shuffle_voxels=False, debug=True)
    layer.eval()
    torch.manual_seed(9)
    net = ref.sst_v1.SSTv1(d_model=[64, 64], nhead=[4, 4], num_blocks=2, dim_feedforward=[128, 128],
                           output_shape=[468, 468],
                           num_attached_conv=0, debug=True, drop_info=(DROP_TRAIN, DROP_TEST), pos_temperature=10000,
                           normalize_pos=False, window_shape=(12, 12))
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn(p_.shape, generator=g) * 0.1)
    net.eval()
    feats = torch.randn(m, 64, generator=g)
    with torch.no_grad():
        vf, ind_list, info = layer(feats, coors.int())
        bev = net((vf, ind_list, info))[0]
    c = info['coors']
    out_feats = bev[c[:, 0], :, c[:, 2], c[:, 3]]
    arrays = {'in::voxel_coors': t2n(coors).astype(np.int32), 'in::voxel_feats': t2n(feats),
              'out::voxel_keep_inds': t2n(info['voxel_keep_inds']), 'out::coors': t2n(c),
              'out::bev_at_voxels': t2n(out_feats), 'out::bev_abs_sum': np.asarray(float(bev.abs().sum()))}
    for s_ in range(2):
        for k in (f'batch_win_inds_shift{s_}', f'coors_in_win_shift{s_}', f'voxel_drop_level_shift{s_}'):
            arrays['out::' + k] = t2n(info[k])
    arrays.update(state_to_np(net.state_dict()))
    print('v1 voxels', m, '->', c.size(0))
    save('sst_v1.npz', **arrays)


def gen_dynamic_vfe(ref):
    g = torch.Generator().manual_seed(4)
    pts_list, coors_list = [], []
    mod = build_ref.load()
    for b in range(2):
        p = torch.rand(700, 3, generator=g) * torch.tensor([7.0, 7.0, 6.0]) + torch.tensor([-3.0 + 40 * b, -3.0, -2.0])
        c = torch.zeros((700, 3), dtype=torch.int32)
        mod.dynamic_voxelize(p.contiguous(), c, list(VOXEL_SIZE), PC_RANGE, 3)
        pts_list.append(p)
        coors_list.append(torch.nn.functional.pad(c, (1, 0), value=b))
    pts, coors = torch.cat(pts_list), torch.cat(coors_list)
    torch.manual_seed(5)
    vfe = ref.voxel_encoder.DynamicVFE(in_channels=3, feat_channels=[64, 128], with_distance=False,
                                       voxel_size=VOXEL_SIZE, with_cluster_center=True, with_voxel_center=True,
                                       point_cloud_range=PC_RANGE,
                                       norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01))
    vfe.train()
    pts.requires_grad_(True)
    vf, vc = vfe(pts, coors)
    gout = torch.randn(vf.shape, generator=g)
    (vf * gout).sum().backward()
    arrays = {'in::points': t2n(pts), 'in::coors': t2n(coors), 'in::grad_out': t2n(gout),
              'out::voxel_feats': t2n(vf), 'out::voxel_coors': t2n(vc), 'out::grad_points': t2n(pts.grad)}
    arrays.update(state_to_np(vfe.state_dict()))
    save('dynamic_vfe.npz', **arrays)


def gen_scatter_vfe(ref):
    """Reference DynamicScatterVFE (FSD segmentor voxel encoder, configs/fsd/fsd_waymoD1_1x.py:32-44)."""
    g = torch.Generator().manual_seed(10)
    vs, rng = (0.25, 0.25, 0.2), [-80, -80, -2, 80, 80, 4]
    mod = build_ref.load()
    pts_list, coors_list = [], []
    for b in range(2):
        p = torch.rand(500, 5, generator=g) * torch.tensor([4.0, 4.0, 3.0, 1, 1]) + torch.tensor([10.0 * b, -2.0, -1.5, 0, 0])
        c = torch.zeros((500, 3), dtype=torch.int32)
        mod.dynamic_voxelize(p.contiguous(), c, list(vs), rng, 3)
        pts_list.append(p)
        coors_list.append(torch.nn.functional.pad(c, (1, 0), value=b))
    pts, coors = torch.cat(pts_list), torch.cat(coors_list).long()
    torch.manual_seed(11)
    vfe = ref.voxel_encoder.DynamicScatterVFE(in_channels=5, feat_channels=[64, 64], voxel_size=vs,
                                              with_cluster_center=True, with_voxel_center=True,
                                              point_cloud_range=rng,
                                              norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01),
                                              unique_once=True)
    vfe.train()
    pts.requires_grad_(True)
    vf, vc, inv = vfe(pts, coors, return_inv=True)
    gout = torch.randn(vf.shape, generator=g)
    (vf * gout).sum().backward()
    arrays = {'in::points': t2n(pts), 'in::coors': t2n(coors), 'in::grad_out': t2n(gout), 'out::voxel_feats': t2n(vf),
              'out::voxel_coors': t2n(vc), 'out::inv': t2n(inv), 'out::grad_points': t2n(pts.grad)}
    arrays.update(state_to_np(vfe.state_dict()))
    save('scatter_vfe.npz', **arrays)


def gen_sir(ref):
    g = torch.Generator().manual_seed(6)
    p = 320
    cluster = torch.randint(0, 24, (p,), generator=g)
    coors = torch.stack([torch.randint(0, 3, (p,), generator=g), torch.randint(0, 2, (p,), generator=g), cluster], 1)
    centers = torch.randn(24, 3, generator=g) * 20
    xyz = centers[cluster] + torch.randn(p, 3, generator=g)
    points = torch.cat([xyz, torch.rand(p, 2, generator=g)], 1)
    features = torch.randn(p, 79, generator=g)
    torch.manual_seed(7)
    sir = ref.sir.SIR(num_blocks=3, in_channels=[84, 133, 133], feat_channels=[[128, 128]] * 3,
                      rel_mlp_hidden_dims=[[16, 32], [16, 32], [16, 32]], norm_cfg=dict(type='LN', eps=1e-3),
                      mode='max', xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)
    sir.train()
    features.requires_grad_(True)
    f_cluster = xyz - centers[cluster]
    pts_feats, cluster_feats, cluster_coors = sir(points, features, coors, f_cluster)
    g1 = torch.randn(pts_feats.shape, generator=g)
    g2 = torch.randn(cluster_feats.shape, generator=g)
    ((pts_feats * g1).sum() + (cluster_feats * g2).sum()).backward()
    arrays = {'in::points': t2n(points), 'in::features': t2n(features), 'in::coors': t2n(coors),
              'in::f_cluster': t2n(f_cluster), 'in::g_pts': t2n(g1), 'in::g_cluster': t2n(g2),
              'out::pts_feats': t2n(pts_feats), 'out::cluster_feats': t2n(cluster_feats),
              'out::cluster_coors': t2n(cluster_coors), 'out::grad_features': t2n(features.grad)}
    arrays.update(state_to_np(sir.state_dict()))
    save('sir.npz', **arrays)


def gen_cluster():
    """FSD cluster assignment: the reference's find_connected_componets / ..._single_batch executed from their own
    source text (detectors/single_stage_fsd.py:45-84; the file itself needs mmdet / mmseg to import)."""
    from scipy.sparse.csgraph import connected_components
    rel = 'mmdet3d/models/detectors/single_stage_fsd.py'
    f_train = ref_loader.load_reference_function(rel, 'find_connected_componets',
                                                 {'connected_components': connected_components})
    f_test = ref_loader.load_reference_function(rel, 'find_connected_componets_single_batch',
                                                {'connected_components': connected_components})
    g = torch.Generator().manual_seed(7)
    arrays = {}
    # (class, connected_dist, spread): the three classes of configs/fsd/fsd_waymoD1_1x.py:281-285
    for tag, dist, sigma, n in (('car', 0.6, 0.7, 1800), ('cyclist', 0.4, 0.35, 900), ('pedestrian', 0.1, 0.06, 1200)):
        centers = torch.rand(60, 2, generator=g) * 100 - 50
        pts = centers[torch.randint(0, 60, (n,), generator=g)] + torch.randn(n, 2, generator=g) * sigma
        pts = torch.cat([pts, torch.rand(n, 1, generator=g) * 4 - 2], 1)
        batch = torch.sort(torch.randint(0, 3, (n,), generator=g))[0].int()
        arrays[f'in::{tag}::points'] = t2n(pts)
        arrays[f'in::{tag}::batch'] = t2n(batch)
        arrays[f'in::{tag}::dist'] = np.float32(dist)
        arrays[f'out::{tag}::train'] = t2n(f_train(pts, batch, dist))
        arrays[f'out::{tag}::test'] = t2n(f_test(pts, batch, dist))
    save('cluster.npz', **arrays)


def gen_point_pool():
    """Dynamic point pool (f3): the membership part is pinned with the reference's own CPU points-in-boxes routine
    (ops/roiaware_pool3d/src/points_in_boxes_cpu.cpp, compiled by oracle/build_ref.build_points_in_boxes) on the
    boxes themselves and on the boxes enlarged by extra_wlh (w, l, h grown, bottom z lowered by half the growth).
    Stored sparsely: the (roi, point) pairs that are inside."""
    mod = build_ref.load_points_in_boxes()
    assert mod is not None, 'build oracle/_ref first (python oracle/build_ref.py)'
    rng = np.random.default_rng(11)
    arrays = {}
    for tag, n_rois, n_pts, extra in (('veh', 150, 12000, (0.5, 0.5, 0.5)), ('ped', 300, 6000, (0.25, 0.5, 1.0))):
        big = tag == 'veh'
        rois = np.concatenate([rng.uniform(-40, 40, (n_rois, 2)), rng.uniform(-2, 1, (n_rois, 1)),
                               rng.uniform(1.5 if big else 0.4, 5.0 if big else 1.2, (n_rois, 3)),
                               rng.uniform(-4, 4, (n_rois, 1))], 1).astype(np.float32)
        pts = np.concatenate([rng.uniform(-45, 45, (n_pts, 2)), rng.uniform(-3, 4, (n_pts, 1))], 1).astype(np.float32)
        near = n_pts * 3 // 4  # most points are scattered around boxes so that many pairs exist
        k = rng.integers(0, n_rois, near)
        pts[:near, :2] = rois[k, :2] + rng.normal(0, 1.2 if big else 0.5, (near, 2)).astype(np.float32)
        pts[:near, 2] = rois[k, 2] + rng.uniform(-0.6, 1.2, near).astype(np.float32) * rois[k, 5]
        e = np.asarray(extra, dtype=np.float32)
        large = rois.copy()
        large[:, 3:6] += e[None, :]
        large[:, 2] -= e[2] * np.float32(0.5)
        pairs = {}
        for name, boxes in (('small', rois), ('large', large)):
            flags = torch.zeros(n_rois, n_pts, dtype=torch.int32)
            mod.points_in_boxes_cpu(torch.from_numpy(np.ascontiguousarray(boxes)), torch.from_numpy(pts), flags)
            pairs[name] = np.stack(np.nonzero(flags.numpy()), 1).astype(np.int32)
        arrays[f'in::{tag}::rois'] = rois
        arrays[f'in::{tag}::pts'] = pts
        arrays[f'in::{tag}::extra_wlh'] = e
        arrays[f'out::{tag}::pairs_in_box'] = pairs['small']
        arrays[f'out::{tag}::pairs_in_enlarged_box'] = pairs['large']
    save('point_pool.npz', **arrays)


SPCONV_CASES = {
    # tag: (n, batch, spatial_shape, ksize, stride, padding, dilation, subm, transpose)
    'subm3': (1500, 2, [12, 40, 40], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False),
    'down3s2': (1500, 2, [12, 40, 40], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, False),
    'down_k313': (1200, 1, [9, 33, 31], [3, 1, 3], [2, 1, 2], [0, 0, 1], [1, 1, 1], False, False),
    'down2s2': (1200, 2, [8, 20, 20], [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], False, False),
    'transposed': (600, 2, [6, 16, 16], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, True),
    'subm_dil2': (1500, 1, [10, 30, 30], [3, 3, 3], [1, 1, 1], [0, 0, 0], [2, 2, 2], True, False),
}


def gen_spconv():
    """Sparse-convolution rulebooks (f4) from the reference's own CPU templates (include/spconv/geometry.h, compiled
    by oracle/build_ref.build_spconv_rulebook): output indices in the CPU path's first-appearance order, pair lists,
    pair counts.  Consumers compare through coordinates (the GPU path and this repo number outputs in sorted order)."""
    mod = build_ref.load_spconv_rulebook()
    assert mod is not None, 'build oracle/_ref first (python oracle/build_ref.py)'
    from oracle import spconv_oracle
    rng = np.random.default_rng(21)
    arrays = {}
    for tag, (n, batch, shape, ks, st, pd, dl, subm, tr) in SPCONV_CASES.items():
        vol = int(np.prod(shape))
        lin = rng.choice(batch * vol, n, replace=False)
        b, r = lin // vol, lin % vol
        ind = np.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]], 1).astype(np.int32)
        if subm:
            out_shape, st_ref, pd_ref = shape, [1, 1, 1], [k // 2 for k in ks]  # spconv_ops.h:74-77
        elif tr:
            out_shape, st_ref, pd_ref = spconv_oracle.deconv_output_size(shape, ks, st, pd, dl, [0, 0, 0]), st, pd
        else:
            out_shape, st_ref, pd_ref = spconv_oracle.conv_output_size(shape, ks, st, pd, dl), st, pd
        outids, pairs, num = mod.get_indice_pairs_3d(torch.from_numpy(ind), batch, out_shape, ks, st_ref, pd_ref, dl,
                                                     subm, tr)
        arrays[f'in::{tag}::indices'] = ind
        # batch, spatial_shape, ksize, stride, padding, dilation, subm, transpose
        arrays[f'in::{tag}::params'] = np.asarray([batch] + shape + ks + st + pd + dl + [int(subm), int(tr)], dtype=np.int32)
        arrays[f'out::{tag}::out_shape'] = np.asarray(out_shape, dtype=np.int32)
        arrays[f'out::{tag}::outids'] = outids.numpy()
        arrays[f'out::{tag}::pairs'] = pairs.numpy()
        arrays[f'out::{tag}::num'] = num.numpy()
        if tag in ('down3s2', 'down2s2', 'subm3'):
            # max pooling over the same pairs with the reference's own CPU functors (src/maxpool.cc); some features are
            # made equal on purpose (ties all receive the gradient) and many are negative (the zero-filled start clips)
            feats = torch.from_numpy(rng.normal(size=(n, 6)).round(1).astype(np.float32))
            gout = torch.from_numpy(rng.normal(size=(len(outids), 6)).astype(np.float32))
            pooled = mod.indice_maxpool(feats, pairs, num, len(outids))
            arrays[f'in::{tag}::pool_features'] = feats.numpy()
            arrays[f'in::{tag}::pool_grad_out'] = gout.numpy()
            arrays[f'out::{tag}::pooled'] = pooled.numpy()
            arrays[f'out::{tag}::pool_grad_in'] = mod.indice_maxpool_backward(feats, pooled, gout, pairs, num).numpy()
    save('spconv.npz', **arrays)


SPARSE_UNET_CFG = dict(in_channels=8, sparse_shape=[16, 40, 40], order=('conv', 'norm', 'act'),
                       norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=32,
                       encoder_channels=((16, ), (16, 16, 16), (32, 32, 32), (32, 32, 32)),
                       encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                       decoder_channels=((32, 32, 32), (32, 32, 16), (16, 16, 16), (16, 16, 16)),
                       decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)))


VOXEL_MIXER_CFG = dict(in_channels=8, sparse_shape=[16, 40, 40], order=('conv', 'norm', 'act'),
                       norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=24,
                       encoder_channels=((16, ), (16, 16), (16, 16)), encoder_paddings=((1, ), (1, 1), (1, 1)),
                       decoder_channels=((16, 16, 16), (16, 16, 16), (16, 16, 16)),
                       decoder_paddings=((1, 1), (1, 1), (1, 1)))


def gen_sparse_unet():
    """FSD's segmentor backbone (f4): the reference's own SimpleSparseUNet (middle_encoders/sparse_unet.py), its
    sparse blocks (ops/sparse_block.py) and its vendored spconv Python package executed unmodified on CPU through
    oracle/ref_loader.load_reference_spconv (rulebook = the reference's compiled CPU templates, convolution
    arithmetic = the per-offset gather / mm / scatter-add in torch).  Training mode (batch statistics) and eval mode
    (perturbed running statistics), outputs and gradients."""
    R = ref_loader.load_reference_spconv()
    torch.manual_seed(5)
    net = R.sparse_unet.SimpleSparseUNet(**SPARSE_UNET_CFG)   # fp32: the reference's ops dispatch on float32 / half
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    rng = np.random.default_rng(31)
    shape, batch, n_cells = SPARSE_UNET_CFG['sparse_shape'], 2, 500
    hs = [shape[0] // 2, shape[1] // 2, shape[2] // 2]
    vol = int(np.prod(hs))
    lin = rng.choice(batch * vol, n_cells, replace=False)
    b, r = lin // vol, lin % vol
    base = np.stack([b, r // (hs[1] * hs[2]), (r // hs[2]) % hs[1], r % hs[2]], 1)
    ind = np.unique(np.concatenate([base * [1, 2, 2, 2] + [0, dz, dy, dx] for dz in (0, 1) for dy in (0, 1)
                                    for dx in (0, 1) if rng.random() < 0.7]), axis=0).astype(np.int32)
    ind = ind[rng.permutation(len(ind))]
    x = torch.randn(len(ind), SPARSE_UNET_CFG['in_channels'])
    gy = torch.randn(len(ind), SPARSE_UNET_CFG['decoder_channels'][-1][-1])
    arrays = {'in::indices': ind, 'in::features': t2n(x.float()), 'in::grad_out': t2n(gy.float())}
    state = {k: v.clone() for k, v in net.state_dict().items()}
    arrays.update(state_to_np({k: v.float() for k, v in state.items()}))
    for mode in ('train', 'eval'):
        net.load_state_dict(state)
        net.train(mode == 'train')
        net.zero_grad()
        xa = x.clone().requires_grad_(True)
        out = net({'voxel_feats': xa, 'voxel_coors': torch.from_numpy(ind)})[0]
        assert torch.equal(out['voxel_coors'], torch.from_numpy(ind))
        (out['voxel_feats'] * gy).sum().backward()
        arrays[f'out::{mode}::voxel_feats'] = t2n(out['voxel_feats'].float())
        arrays[f'out::{mode}::grad_features'] = t2n(xa.grad.float())
        for name in ('conv_input.0.weight', 'encoder_layers.encoder_layer3.0.0.weight', 'lateral_layer2.conv2.weight',
                     'upsample_layer3.0.weight', 'merge_layer1.1.weight'):
            arrays[f'out::{mode}::grad::{name}'] = t2n(dict(net.named_parameters())[name].grad.float())
    save('sparse_unet.npz', **arrays)
    # FSDv2's VirtualVoxelMixer (sparse_unet.py:417-504) in the shape of configs/fsdv2/fsdv2_waymo_1x.py:127-139, narrower
    torch.manual_seed(6)
    mix = R.sparse_unet.VirtualVoxelMixer(**VOXEL_MIXER_CFG)
    mix.train()
    xa = x.clone().requires_grad_(True)
    feats, out_ind, out_shape = mix(xa, torch.from_numpy(ind), batch)
    assert torch.equal(out_ind, torch.from_numpy(ind))
    gy2 = torch.randn(feats.shape)
    (feats * gy2).sum().backward()
    arrays2 = {'in::indices': ind, 'in::features': t2n(x), 'in::grad_out': t2n(gy2), 'out::features': t2n(feats),
               'out::grad_features': t2n(xa.grad),
               'out::grad::conv_out.0.weight': t2n(mix.conv_out[0].weight.grad),
               'out::grad::encoder_layers.encoder_layer2.0.0.weight':
                   t2n(mix.encoder_layers.encoder_layer2[0][0].weight.grad)}
    arrays2.update(state_to_np(mix.state_dict()))
    save('voxel_mixer.npz', **arrays2)


VIRTUAL_VOXEL_CFG = dict(
    virtual_point_projector=dict(in_channels=16 + 3 + 4 + 2, hidden_dims=[16, 16], norm_cfg=dict(type='naiveSyncBN1d'),
                                 ori_in_channels=16, ori_hidden_dims=[16, 16]),
    voxel_encoder=dict(type='DynamicScatterVFE', in_channels=3 + 16, feat_channels=[16, 8], voxel_size=(0.4, 0.4, 0.4),
                       with_cluster_center=True, with_voxel_center=True, point_cloud_range=[-8, -8, -3.2, 8, 8, 3.2],
                       norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), unique_once=True),
    backbone=dict(type='VirtualVoxelMixer', **VOXEL_MIXER_CFG))


VIRTUAL_VOXEL_MS_CFG = dict(
    VIRTUAL_VOXEL_CFG,
    virtual_point_projector=dict(VIRTUAL_VOXEL_CFG['virtual_point_projector'], recover_in_channels=24 + 3,
                                 recover_hidden_dims=[16, 16]),
    multiscale_cfg=dict(multiscale_levels=[0, 1], projector_hiddens=[[12, 8], [8, 16, 8]], fusion_mode='avg',
                        target_sparse_shape=[16, 40, 40], norm_cfg=dict(type='naiveSyncBN1d')))


def gen_virtual_voxel(multiscale=False):
    """FSDv2's virtual-voxel chain: SingleStageFSDV2.extract_feat (single_stage_fsd_v2.py:159-271, non-baseline mode,
    training), the method's own source executed on a stand-in ``self`` that carries the reference's own submodules
    (build_mlp projectors, DynamicScatterVFE, VirtualVoxelMixer over the vendored spconv package) on CPU.
    ``multiscale``: the second fixture (virtual_voxel_ms.npz) - the same inputs plus two coarser levels of "decoder features"
    (objects with features / indices / spatial_shape, as the segmentor's SparseConvTensors) fused in by the reference's own
    multiscale_fusion / ms_coors_proj (:375-433), and ``as_rpn`` on: recover_point_features and the pts_* outputs (:131-155,
    263-270)."""
    import types
    R = ref_loader.load_reference_spconv()
    ref = ref_loader.load_reference()
    rel = 'mmdet3d/models/detectors/single_stage_fsd_v2.py'
    glb = {'scatter_v2': ref.sst_ops.scatter_v2}
    cfg = VIRTUAL_VOXEL_MS_CFG if multiscale else VIRTUAL_VOXEL_CFG
    vpp = cfg['virtual_point_projector']
    torch.manual_seed(21)
    me = types.SimpleNamespace()
    me.baseline_mode, me.zero_virtual_feature, me.only_virtual, me.training, me.as_rpn = False, False, False, True, multiscale
    me.train_cfg, me.print_info = {}, {}
    me.virtual_voxel_size, me.point_cloud_range = cfg['voxel_encoder']['voxel_size'], cfg['voxel_encoder']['point_cloud_range']
    me.virtual_proj = ref.sst_ops.build_mlp(vpp['in_channels'], vpp['hidden_dims'], vpp['norm_cfg'])
    me.ori_proj = ref.sst_ops.build_mlp(vpp['ori_in_channels'], vpp['ori_hidden_dims'], vpp['norm_cfg'])
    vfe_cfg = dict(cfg['voxel_encoder'])
    vfe_cfg.pop('type')
    me.voxel_encoder = ref.voxel_encoder.DynamicScatterVFE(**vfe_cfg)
    me.backbone = R.sparse_unet.VirtualVoxelMixer(**VOXEL_MIXER_CFG)
    mods = [('virtual_proj', me.virtual_proj), ('ori_proj', me.ori_proj), ('voxel_encoder', me.voxel_encoder),
            ('backbone', me.backbone)]
    methods = ['voxelize_with_batch_idx', 'clip_points']
    if multiscale:
        ms_cfg = me.multiscale_cfg = cfg['multiscale_cfg']
        me.recover_proj = ref.sst_ops.build_mlp(vpp['recover_in_channels'], vpp['recover_hidden_dims'], vpp['norm_cfg'])
        me.ms_projectors = torch.nn.ModuleList([ref.sst_ops.build_mlp(p[0], p[1:], ms_cfg['norm_cfg'])
                                                for p in ms_cfg['projector_hiddens']])
        mods += [('recover_proj', me.recover_proj), ('ms_projectors', me.ms_projectors)]
        methods += ['multiscale_fusion', 'ms_coors_proj', 'recover_point_features']
    for _, m in mods:
        m.train()
    for fn in methods:
        f = ref_loader.load_reference_method(rel, 'SingleStageFSDV2', fn, glb)
        setattr(me, fn, types.MethodType(f, me))
    extract = ref_loader.load_reference_method(rel, 'SingleStageFSDV2', 'extract_feat', glb)
    g = torch.Generator().manual_seed(22)
    n_ori, n_fg, batch = 900, 260, 2
    rng_lo, rng_hi = torch.tensor([-8.0, -8.0, -3.2]), torch.tensor([8.0, 8.0, 3.2])
    ori_xyz = torch.rand(n_ori, 3, generator=g) * (rng_hi - rng_lo) * 0.6 + rng_lo * 0.6
    ori = dict(seg_points=torch.cat([ori_xyz, torch.rand(n_ori, 2, generator=g)], 1),
               seg_feats=torch.randn(n_ori, 16, generator=g), batch_idx=torch.randint(0, batch, (n_ori,), generator=g))
    sel = torch.randperm(n_ori, generator=g)[:n_fg]
    centers = ori_xyz[sel] + torch.randn(n_fg, 3, generator=g) * 0.8
    centers[:6] += 20.0                                   # some predicted centres outside the range: clip_points
    smp = dict(seg_points=ori['seg_points'][sel].clone(), center_preds=centers, seg_logits=torch.randn(n_fg, 4, generator=g),
               seg_feats=ori['seg_feats'][sel].clone(), batch_idx=ori['batch_idx'][sel].clone())
    leaves = {k: v.clone().requires_grad_(True) for k, v in (('ori_feats', ori['seg_feats']), ('smp_feats', smp['seg_feats']),
                                                             ('smp_logits', smp['seg_logits']))}
    ori_in = dict(ori, seg_feats=leaves['ori_feats'])
    smp_in = dict(smp, seg_feats=leaves['smp_feats'], seg_logits=leaves['smp_logits'], center_preds=centers.clone())
    arrays = {}
    ms_features = None
    if multiscale:
        # two coarser levels: distinct random cells of a [8, 20, 20] grid (strides 2, 2 against the [16, 40, 40] target) and
        # of a [16, 40, 40] grid (strides 1, 1: lands ON virtual voxels), int32 (b, z, y, x) like SparseConvTensor.indices
        ms_features = []
        for lvl, (shape, n_vox, width) in enumerate((([8, 20, 20], 500, 12), ([16, 40, 40], 700, 8))):
            cells = torch.randperm(batch * shape[0] * shape[1] * shape[2], generator=g)[:n_vox].sort()[0]
            b, rem = cells // (shape[0] * shape[1] * shape[2]), cells % (shape[0] * shape[1] * shape[2])
            ind = torch.stack([b, rem // (shape[1] * shape[2]), rem // shape[2] % shape[1], rem % shape[2]], 1).int()
            feats = torch.randn(n_vox, width, generator=g).requires_grad_(True)
            ms_features.append(types.SimpleNamespace(features=feats, indices=ind, spatial_shape=shape))
            arrays[f'in::ms{lvl}::features'], arrays[f'in::ms{lvl}::indices'] = t2n(feats), t2n(ind)
            arrays[f'in::ms{lvl}::spatial_shape'] = np.asarray(shape)
    out = extract(me, smp_in, ori_in, None, ms_features)
    gy = torch.randn(out['virtual_feats'].shape, generator=g)
    loss = (out['virtual_feats'] * gy).sum()
    if multiscale:
        gp = torch.randn(out['pts_feats'].shape, generator=g)
        loss = loss + (out['pts_feats'] * gp).sum()
        arrays['in::grad_pts'] = t2n(gp)
    loss.backward()
    arrays.update({'in::ori_points': t2n(ori['seg_points']), 'in::ori_feats': t2n(ori['seg_feats']),
                   'in::ori_batch_idx': t2n(ori['batch_idx']), 'in::smp_points': t2n(smp['seg_points']),
                   'in::smp_centers': t2n(centers), 'in::smp_logits': t2n(smp['seg_logits']), 'in::smp_feats': t2n(smp['seg_feats']),
                   'in::smp_batch_idx': t2n(smp['batch_idx']), 'in::grad_out': t2n(gy),
                   'out::virtual_feats': t2n(out['virtual_feats']), 'out::virtual_coors': t2n(out['virtual_coors']),
                   'out::virtual_centers': t2n(out['virtual_centers']), 'out::virtual_centroid': t2n(out['virtual_centroid']),
                   'out::sparse_shape': np.asarray(out['sparse_shape']),
                   'out::grad_ori_feats': t2n(leaves['ori_feats'].grad), 'out::grad_smp_feats': t2n(leaves['smp_feats'].grad),
                   'out::grad_smp_logits': t2n(leaves['smp_logits'].grad)})
    for prefix, mod in mods:
        arrays.update({f'w::{prefix}.{k}': t2n(v.float()) for k, v in mod.state_dict().items()})
    arrays['out::grad::virtual_proj.0.0.weight'] = t2n(me.virtual_proj[0][0].weight.grad)
    arrays['out::grad::ori_proj.1.0.weight'] = t2n(me.ori_proj[1][0].weight.grad)
    if multiscale:
        for key in ('pts_feats', 'pts_xyz', 'pts_indicators', 'pts_batch_inds'):
            arrays['out::' + key] = t2n(out[key])
        for lvl, d in enumerate(ms_features):
            arrays[f'out::grad_ms{lvl}'] = t2n(d.features.grad)
        arrays['out::grad::ms_projectors.0.0.0.weight'] = t2n(me.ms_projectors[0][0][0].weight.grad)
        arrays['out::grad::ms_projectors.1.1.0.weight'] = t2n(me.ms_projectors[1][1][0].weight.grad)
        arrays['out::grad::recover_proj.0.0.weight'] = t2n(me.recover_proj[0][0].weight.grad)
    save('virtual_voxel_ms.npz' if multiscale else 'virtual_voxel.npz', **arrays)


CHAIN_GRAD_KEYS = {
    'fsd': ['voxel_encoder.vfe_layers.0.linear.weight', 'seg_backbone.conv_input.0.weight',
            'seg_backbone.encoder_layers.encoder_layer2.0.0.weight', 'seg_backbone.upsample_layer2.0.weight',
            'seg_backbone.lateral_layer1.conv2.weight', 'seg_head.weight', 'backbone.block_list.0.rel_mlp.0.0.weight',
            'backbone.block_list.1.vfe_layers.1.linear.weight', 'backbone.block_list.2.vfe_layers.0.norm.weight'],
    'fsdv2': ['voxel_encoder.vfe_layers.1.linear.weight', 'seg_backbone.encoder_layers.encoder_layer3.0.0.weight',
              'seg_backbone.merge_layer1.0.weight', 'seg_head.weight', 'virtual_stage.virtual_proj.0.0.weight',
              'virtual_stage.ori_proj.1.0.weight', 'virtual_stage.voxel_encoder.vfe_layers.0.linear.weight',
              'virtual_stage.backbone.conv_out.0.weight', 'virtual_stage.backbone.encoder_layers.encoder_layer2.0.0.weight',
              'virtual_stage.ms_projectors.0.0.0.weight', 'virtual_stage.ms_projectors.1.0.0.weight',
              'virtual_stage.recover_proj.0.0.weight', 'seg_backbone.upsample_layer4.0.weight'],
}


def gen_fsd_chains():
    """BASELINE.json configs[3] / configs[4] as CHAINS, produced by the reference's own modules wired the way its detectors
    wire them (oracle/ref_fsd.reference_ops: DynamicScatterVFE -> SimpleSparseUNet -> Voxel2PointScatterNeck -> ClusterAssigner
    -> SingleStageFSD.extract_feat / SIR, and -> SingleStageFSDV2.extract_feat / VirtualVoxelMixer), at fixture size
    (bench_workloads.FSD_SMALL_CFG / FSDV2_SMALL_CFG), training mode, forward + backward.  The wiring code itself is
    bench_workloads.FSDPath / FSDv2Path - the same code the GPU path and the CPU port run."""
    import bench_workloads as BW
    from oracle import ref_fsd
    ops = ref_fsd.reference_ops()
    for tag, cls, cfg, kw, half in (('fsd', BW.FSDPath, BW.FSD_SMALL_CFG, dict(roi_stage=False), 18.0),
                                    ('fsdv2', BW.FSDv2Path, BW.FSDV2_SMALL_CFG, dict(), 12.0)):
        torch.manual_seed(40)
        net = cls(ops, cfg, **kw).train()
        g = torch.Generator().manual_seed(41)
        with torch.no_grad():
            for p in net.parameters():      # biases and norm parameters away from their 0 / 1 defaults
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.1)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        clouds = [BW.chain_cloud(4000, 1, half_extent=half), BW.chain_cloud(3000, 2, half_extent=half)]
        loss, stats, out = net(clouds, return_tensors=True)
        loss.backward()
        arrays = {f'in::points{i}': t2n(c) for i, c in enumerate(clouds)}
        arrays['out::loss'] = t2n(loss)
        for k, v in out.items():
            if v is None or k in ('seg_feats', 'voxel_feats'):      # implied by unet_feats / head; keeps the fixture small
                continue
            v = v[::4] if k == 'head' else v                        # every 4th point
            arrays['out::' + k] = t2n(v) if v.is_floating_point() else t2n(v).astype(np.int32)
        arrays.update({'stats::' + k: np.asarray(v) for k, v in stats.items()})
        params = dict(net.named_parameters())
        for k in CHAIN_GRAD_KEYS[tag]:
            arrays['grad::' + k] = t2n(params[k].grad)
        # the adjudicator for the gradients: the CPU port of the same chain evaluated in float64 on the same weights and
        # clouds (integer stages in fp32).  The reference's own fp32 backward is up to ~7e-3 away from it at this size
        # (naiveSyncBN differentiates var = E[x^2] - mean^2 in fp32); tests compare against grad64 with a bar that knows that.
        from oracle import fsd_cpu
        port64 = cls(fsd_cpu, cfg, **kw).double().train()
        port64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in state.items()}, strict=True)
        loss64, stats64 = port64([c.double() for c in clouds])
        assert {k: int(v) for k, v in stats64.items()} == {k: int(v) for k, v in stats.items()}
        loss64.backward()
        params64 = dict(port64.named_parameters())
        for k in CHAIN_GRAD_KEYS[tag]:
            arrays['grad64::' + k] = params64[k].grad.numpy().astype(np.float64)
            print('   ', k, 'reference fp32 vs float64: %.2e' % (np.abs(arrays['grad::' + k] - arrays['grad64::' + k]).max()
                                                                 / max(1.0, np.abs(arrays['grad64::' + k]).max())))
        arrays.update(state_to_np(state))
        print(tag, stats)
        save(f'{tag}_chain.npz', **arrays)


def gen_hard_voxelize():
    """Hard voxelization (max_num_points / max_voxels set) from the reference's own compiled C++
    (voxelization_cpu.cpp:43-142 through oracle/_ref/voxel_layer_ref.so): more voxels than max_voxels (later ones
    dropped), more points per voxel than max_points, points outside the range (clamped in this fork)."""
    mod = build_ref.load()
    assert mod is not None
    g = torch.Generator().manual_seed(17)
    arrays = {}
    for tag, n, vs, rng, max_points, max_voxels in (('dense', 6000, [0.5, 0.5, 4.0], [0, -10, -3, 20, 10, 1], 5, 800),
                                                    ('all_kept', 3000, [0.32, 0.32, 6.0], [-10, -10, -2, 10, 10, 4], 32, 20000)):
        lo = torch.tensor(rng[:3]) - 1.0
        hi = torch.tensor(rng[3:]) + 1.0
        pts = torch.rand(n, 4, generator=g) * torch.cat([hi - lo, torch.ones(1)]) + torch.cat([lo, torch.zeros(1)])
        voxels = torch.zeros(max_voxels, max_points, 4)
        coors = torch.zeros(max_voxels, 3, dtype=torch.int32)
        num = torch.zeros(max_voxels, dtype=torch.int32)
        nv = mod.hard_voxelize(pts, voxels, coors, num, vs, rng, max_points, max_voxels, 3)
        arrays[f'in::{tag}::points'] = t2n(pts)
        arrays[f'in::{tag}::params'] = np.asarray(vs + rng + [max_points, max_voxels], dtype=np.float64)
        arrays[f'out::{tag}::voxels'] = t2n(voxels[:nv])
        arrays[f'out::{tag}::coors'] = t2n(coors[:nv])
        arrays[f'out::{tag}::num_points'] = t2n(num[:nv])
    save('hard_voxelize.npz', **arrays)


def main():
    assert ref_loader.available(), 'the reference tree is required'
    if len(sys.argv) > 1 and sys.argv[1] == 'chains':      # only the chain goldens
        build_ref.build()
        gen_fsd_chains()
        return
    build_ref.build()
    ref = ref_loader.load_reference()
    if len(sys.argv) > 2 and sys.argv[1] == 'sst_block':     # regenerate single block variants only
        gen_sst_block(ref, only=sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'sst_block_bf16':
        gen_sst_block_bf16(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'virtual_voxel':
        gen_virtual_voxel()
        gen_virtual_voxel(multiscale=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'virtual_voxel_ms':
        gen_virtual_voxel(multiscale=True)
        return
    gen_voxelize()
    gen_hard_voxelize()
    gen_input_layer(ref)
    gen_sst_block(ref)
    gen_sst_block_bf16(ref)
    gen_sst_bev(ref)
    gen_sst_v1(ref)
    gen_dynamic_vfe(ref)
    gen_scatter_vfe(ref)
    gen_sir(ref)
    gen_cluster()
    build_ref.build_points_in_boxes()
    gen_point_pool()
    build_ref.build_spconv_rulebook()
    gen_spconv()
    gen_sparse_unet()
    gen_virtual_voxel()
    gen_virtual_voxel(multiscale=True)
    gen_fsd_chains()


if __name__ == '__main__':
    main()
