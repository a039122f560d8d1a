"""GPU: the fused index plan (csrc/frame_plan.hip: no host round trips) against the piecewise path through the module
interfaces (Voxelization -> DynamicVFE.scatter_plan -> SSTInputLayerV2.build_plan), which itself is pinned to the
reference's golden tensors (tests/test_gpu_window.py, test_gpu_voxel.py).  Integer outputs bit-exact."""
import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN, PC_RANGE, VOXEL_SIZE

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _modules(shuffle=False, train=True, window_shape=(12, 12, 1)):
    import sst_amd
    vox = sst_amd.Voxelization(VOXEL_SIZE, PC_RANGE, -1, (-1, -1))
    vfe = sst_amd.DynamicVFE(in_channels=3, feat_channels=[64, 128], voxel_size=VOXEL_SIZE, with_cluster_center=True,
                             with_voxel_center=True, point_cloud_range=PC_RANGE,
                             norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)).to(DEV)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), window_shape, (468, 468, 1), shuffle_voxels=shuffle,
                                    debug=False, mute=True, reference_outputs=False, window_major=True)
    layer.train(train)
    return vox, vfe, layer


def _clouds(kind):
    g = torch.Generator().manual_seed(7)
    span = torch.tensor([149.76, 149.76, 6.0])
    lo = torch.tensor([-74.88, -74.88, -2.0])
    if kind == 'uniform':
        return [torch.rand(20000, 3, generator=g) * span + lo]
    if kind == 'crowded2':     # two samples, windows far above the 100-token cap, points outside the range
        a = torch.cat([torch.rand(9000, 3, generator=g) * span + lo,
                       torch.rand(9000, 3, generator=g) * torch.tensor([9.0, 9.0, 6.0]) + torch.tensor([3.0, -20.0, -2.0]),
                       torch.rand(200, 3, generator=g) * 400 - 200])
        b = torch.cat([torch.rand(4000, 3, generator=g) * span + lo,
                       torch.rand(6000, 3, generator=g) * torch.tensor([6.0, 12.0, 6.0]) + torch.tensor([-60.0, 40.0, -2.0])])
        return [a, b]
    if kind == 'tiny3':
        return [torch.rand(5, 3, generator=g) * span + lo, torch.rand(1, 3, generator=g) * span + lo,
                torch.rand(40, 3, generator=g) * span + lo]
    raise KeyError(kind)


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('kind', ['uniform', 'crowded2', 'tiny3'])
def test_fused_plan_equals_piecewise_path(kind, train):
    from sst_amd.frame_plan import FramePlanner
    vox, vfe, layer = _modules(train=train)
    clouds = [c.to(DEV) for c in _clouds(kind)]
    # piecewise
    points, coors = vox.voxelize_batch(clouds)
    sp = vfe.scatter_plan(coors)
    wplan = layer.build_plan(sp.voxel_coors, len(clouds), 128, torch.float32)
    # fused
    planner = FramePlanner(vox, vfe, layer)
    assert planner.supported(len(clouds))
    plan = planner.build(clouds)
    n = coors.size(0)
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(n, 8, generator=g).to(DEV)
    red_f = {mode: plan.reduce(feats, mode) for mode in ('max', 'mean')}
    vfeat = torch.randn(plan.n_upper, 128, generator=g).to(DEV)
    info_f = plan.finalize(vfeat, layer)
    m = plan.num_voxels
    assert m == sp.num_voxels
    assert torch.equal(plan.coors_map, sp.coors_map)
    assert torch.equal(plan.vcoors[:m], sp.voxel_coors)
    for mode in ('max', 'mean'):
        assert torch.equal(red_f[mode][:m], sp.reduce(feats, mode))
    info_p = layer.apply_plan(wplan, vfeat[:m])
    assert torch.equal(info_f['voxel_coors'], info_p['voxel_coors'])
    assert torch.equal(info_f['voxel_keep_inds'], info_p['voxel_keep_inds'])
    assert torch.equal(info_f['voxel_feats'], info_p['voxel_feats'])
    if kind == 'crowded2' and train:
        assert info_f['voxel_feats'].size(0) < m, 'the crowded cloud must lose voxels to the drop'
    for i in range(2):
        pf, pp = info_f[f'sra_plan_shift{i}'], info_p[f'sra_plan_shift{i}']
        assert (pf.n_windows, pf.n_tokens, pf.max_tokens) == (pp.n_windows, pp.n_tokens, pp.max_tokens)
        assert torch.equal(pf.winoff[:pf.n_windows + 1], pp.winoff[:pp.n_windows + 1])
        assert torch.equal(pf.tok[:pf.n_tokens], pp.tok[:pp.n_tokens])
        assert torch.equal(info_f[f'pos_embed_shift{i}'], info_p[f'pos_embed_shift{i}'])


def test_fused_plan_gradient_paths():
    """segmented max / mean over the upper-bound sized voxel table and the row gather of finalize(): gradients equal to
    the piecewise path's."""
    from sst_amd.frame_plan import FramePlanner
    vox, vfe, layer = _modules()
    clouds = [c.to(DEV) for c in _clouds('crowded2')]
    points, coors = vox.voxelize_batch(clouds)
    sp = vfe.scatter_plan(coors)
    plan = FramePlanner(vox, vfe, layer).build(clouds)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(coors.size(0), 16, generator=g)
    for mode in ('max', 'mean'):
        fa, fb = feats.to(DEV).requires_grad_(True), feats.to(DEV).requires_grad_(True)
        ra, rb = plan.reduce(fa, mode), sp.reduce(fb, mode)
        w = torch.randn(rb.shape, generator=g).to(DEV)
        m = rb.size(0)
        (ra[:m] * w).sum().backward()
        (rb * w).sum().backward()
        assert torch.equal(fa.grad, fb.grad)
    vfeat = torch.randn(plan.n_upper, 128, generator=g).to(DEV).requires_grad_(True)
    info = plan.finalize(vfeat, layer)
    w = torch.randn(info['voxel_feats'].shape, generator=g).to(DEV)
    (info['voxel_feats'] * w).sum().backward()
    ref = torch.zeros_like(vfeat)
    ref[info['voxel_keep_inds']] = w
    assert torch.equal(vfeat.grad, ref)


def test_fused_plan_random_drop_invariants():
    """shuffle_voxels=True: the survivors of an over-full window are a random subset - every window within its cap,
    window membership consistent with the coordinates, all tokens distinct, another seed another subset."""
    from sst_amd.frame_plan import FramePlanner
    vox, vfe, layer = _modules(shuffle=True)
    clouds = [c.to(DEV) for c in _clouds('crowded2')]
    planner = FramePlanner(vox, vfe, layer)
    kept_sets = []
    for seed in (1, 2):
        torch.manual_seed(seed)
        plan = planner.build(clouds)
        vfeat = torch.zeros(plan.n_upper, 128, device=DEV)
        info = plan.finalize(vfeat, layer)
        coors = info['voxel_coors'].cpu().numpy()
        mk = coors.shape[0]
        assert mk < plan.num_voxels
        key = ((coors[:, 0] * 2 + coors[:, 1]) * 468 + coors[:, 2]) * 468 + coors[:, 3]
        assert np.unique(key).size == mk
        kept_sets.append(set(key.tolist()))
        for s, shift in ((0, 12), (1, 6)):
            p = info[f'sra_plan_shift{s}']
            off = p.winoff[:p.n_windows + 1].cpu().numpy()
            tok = p.tok[:p.n_tokens].cpu().numpy()
            sizes = np.diff(off)
            assert sizes.min() >= 1 and sizes.max() <= 100 and off[-1] == mk
            assert np.array_equal(np.sort(tok), np.arange(mk))
            wid = (coors[:, 0] * 40 + (coors[:, 3] + shift) // 12) * 40 + (coors[:, 2] + shift) // 12
            wid_tok = wid[tok]
            starts = off[:-1]
            seg = np.repeat(np.arange(p.n_windows), sizes)
            assert np.array_equal(wid_tok, wid_tok[starts][seg]), 'a window of the CSR mixes voxels of two windows'
            assert np.all(np.diff(wid_tok[starts]) > 0), 'windows not in ascending id order'
    assert kept_sets[0] != kept_sets[1]


def test_a_steps_index_plan_dies_with_the_step_without_the_cyclic_collector():
    """plan -> voxel_info -> (deferred entries) must not lead back to the plan: with a reference cycle every step's plan (~65 MB of
    index buffers at the bench size) waited for Python's cyclic collector, the caching allocator grew by a few segments per step
    and device memory in use by 65 MB per step (round 5: found by bench.py's allocation trace)"""
    import gc
    import weakref
    import bench
    torch.manual_seed(0)
    model = bench.Pipeline(model_cfg=bench.load_config_fixture('sst_waymoD5_1x_3class_8heads_v2'), voxel_feats_only=True).to(DEV).train()
    model.middle_encoder.mute = True
    frames = [bench.make_cloud(20000, 5, DEV)]
    gc.collect()
    gc.disable()
    try:
        plan = model.prepare(frames)
        assert plan is not None
        out = model(frames, plan)
        out.sum().backward()
        ref = weakref.ref(plan)
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        del plan, out
        model.last_voxel_coors = model.last_plans = None
        model.zero_grad(set_to_none=True)
        assert ref() is None, 'the frame plan is kept alive by a reference cycle'
        assert torch.cuda.memory_allocated() < before
    finally:
        gc.enable()


def test_a_plan_built_ahead_keeps_its_own_sizes_however_long_it_waits():
    """the pinned size words of a plan belong to it until finalize() has read them: a plan built ahead (bench.py keeps one across
    its legs) and finalized after MANY other plans were built must still describe its own frame.  (Round 5: a first version of
    the pinned free list recycled slots round-robin after 16 builds - a plan that waited got another frame's sizes and the
    attention kernels read out of bounds.)"""
    import bench
    torch.manual_seed(0)
    model = bench.Pipeline(1).to(DEV).train()
    big = [bench.make_cloud(40000, 5, DEV)]
    small = [bench.make_cloud(3000, 6, DEV)]
    with torch.no_grad():
        want = model(big).size(0)
        waiting = model.prepare(big)
        for _ in range(40):
            assert model(small).size(0) < want
        out = model(big, waiting)
    assert out.size(0) == want
    assert len(model._planner._count_slots) <= 4, 'slots are handed back: the free list stays small'


def _plan_tensors(plan):
    out = {k: v for k, v in vars(plan).items() if isinstance(v, torch.Tensor) and v.is_cuda}
    out.update({'groups.' + a: getattr(plan.groups, a) for a in ('perm', 'inverse', 'offsets', 'ukeys', 'num')})
    return out


@pytest.mark.parametrize('kind', ['uniform', 'crowded2'])
def test_a_plan_built_beside_the_step_in_flight_equals_the_plan_built_in_line(kind):
    """FramePlanner.build_overlapped (the planner's own stream, while the caller's stream is busy) == build(): every index
    tensor and every size, with a long queue of work on the caller's stream in front of it; the pinned host batch of a data
    loader (DynamicVoxelNet.prepare(host tensors, overlap=True)) gives the same plan again."""
    from sst_amd.frame_plan import FramePlanner
    vox, vfe, layer = _modules()
    host = _clouds(kind)
    clouds = [c.to(DEV) for c in host]
    planner = FramePlanner(vox, vfe, layer)
    want = planner.build(clouds)
    torch.cuda.synchronize()
    a = torch.randn(4096, 4096, device=DEV)
    for _ in range(20):                      # the step in flight: ~tens of ms on the caller's stream
        a = torch.tanh(a @ a * 1e-2)
    got = planner.build_overlapped(clouds)
    assert got.overlapped
    got.counts_ready.synchronize()
    sizes_w, sizes_g = want.h_counts.tolist()[:6], got.h_counts.tolist()[:6]
    assert sizes_w == sizes_g
    wt, gt = _plan_tensors(want), _plan_tensors(got)
    assert set(wt) == set(gt)
    m, m_keep = sizes_w[0], sizes_w[1]
    n_groups = int(want.groups.num.item())
    assert n_groups == int(got.groups.num.item())
    for k in wt:
        n = {'vcoors': m, 'gidx': None, 'feat_index': m_keep, 'feat_index_i32': m_keep, 'out_coors': m_keep, 'tok1': m_keep,
             'posidx0': m_keep, 'posidx1': m_keep, 'groups.ukeys': n_groups, 'groups.offsets': n_groups + 1}.get(k)
        x, y = (wt[k], gt[k]) if n is None else (wt[k][:n], gt[k][:n])
        if k in ('winoff0', 'winoff1'):
            nw = sizes_w[2 + int(k[-1])]
            x, y = x[:nw + 1], y[:nw + 1]
        if k == 'd_counts':
            x, y = x[:6], y[:6]
        if k == 'gidx':
            continue                         # rows of dropped groups are unspecified beyond the counts
        assert torch.equal(x, y), k
    pinned = [c.pin_memory() for c in host]
    plan_h = planner.build_overlapped_from_host(pinned, torch.device(DEV))
    plan_h.counts_ready.synchronize()
    assert plan_h.h_counts.tolist()[:6] == sizes_w
    for k in ('feat_index_i32', 'tok1', 'posidx0', 'posidx1'):
        assert torch.equal(_plan_tensors(plan_h)[k][:m_keep], wt[k][:m_keep]), k
    assert all(torch.equal(p.cpu(), h) for p, h in zip(plan_h.points_list, host))


def test_training_steps_with_the_plan_built_beside_the_backward_pass_are_the_same_steps():
    """twelve forward + backward steps of the pipeline, the next step's plan built (a) in line, (b) ahead on the same stream,
    (c) ahead on the planner's own stream beside the backward pass: outputs and parameter gradients bit for bit equal
    (shuffle off: the drop is deterministic) - the overlapped plan's buffers are never reused while the step still reads them"""
    import bench
    outs = {}
    clouds = [[bench.make_cloud(30000 + 1500 * i, 40 + i, DEV)] for i in range(4)]
    for mode in ('inline', 'same_stream', 'overlap'):
        torch.manual_seed(0)
        cfg = bench._pipeline_config(2)
        cfg['middle_encoder'] = dict(cfg['middle_encoder'], shuffle_voxels=False)
        model = bench.Pipeline(model_cfg=cfg, voxel_feats_only=True).to(DEV).train()
        params = [p for p in model.parameters() if p.requires_grad]
        ahead, rec = [], []
        for step in range(12):
            frames = clouds[step % 4]
            for p in params:
                p.grad = None
            out = model(frames, ahead.pop() if ahead else None)
            g = torch.Generator(device=DEV).manual_seed(step)
            out.backward(torch.randn(out.shape, device=DEV, generator=g))
            if mode != 'inline':
                ahead.append(model.prepare(clouds[(step + 1) % 4], overlap=(mode == 'overlap')))
            rec.append((out.detach().clone(), [p.grad.clone() for p in params]))
            del out
        torch.cuda.synchronize()
        outs[mode] = rec
    for mode in ('same_stream', 'overlap'):
        for (oa, ga), (ob, gb) in zip(outs['inline'], outs[mode]):
            assert torch.equal(oa, ob), mode
            assert all(torch.equal(x, y) for x, y in zip(ga, gb)), mode
