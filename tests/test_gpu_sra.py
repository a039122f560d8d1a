"""GPU: Sparse Regional Attention core (MFMA and generic kernels), forward and backward, against the float64
oracle; encoder layers / SSTv2 block against golden tensors from the reference's own Python.
Tolerance: 1e-3 absolute on fp32 features (BASELINE.json north_star), in practice ~1e-5."""

import os

import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-3


def _plan_from_sizes(sizes, seed):
    from sst_amd import kernels as K
    rng = np.random.default_rng(seed)
    m = int(sum(sizes))
    tok = rng.permutation(m).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    plan = K.WindowPlan(torch.from_numpy(tok).to(DEV), torch.from_numpy(off).to(DEV), len(sizes), m, max(sizes))
    return plan, tok, off, m


SIZE_SETS = {
    'tiny': [1, 2, 3, 15, 16, 17],
    'levels': [30, 31, 32, 33, 47, 48, 49, 60, 63, 64, 65],
    'big': [96, 100, 111, 112, 113, 128, 143, 144],
    'mixed': [1, 144, 7, 100, 64, 30, 16, 59, 81, 12, 5, 133],
    'over_cap': [150, 20, 200, 144, 145],
}


@pytest.mark.parametrize('impl', [0, 1, 2, 3])
@pytest.mark.parametrize('heads', [8, 12])
@pytest.mark.parametrize('name', list(SIZE_SETS))
def test_sra_core_forward_backward_vs_oracle(name, heads, impl):
    from sst_amd import kernels as K
    from oracle import sst_oracle
    sizes = SIZE_SETS[name]
    plan, tok, off, m = _plan_from_sizes(sizes, len(sizes) + heads)
    c = heads * 16
    g = torch.Generator().manual_seed(m + heads)
    q = torch.randn(m, c, generator=g) * 1.5
    k = torch.randn(m, c, generator=g) * 1.5
    v = torch.randn(m, c, generator=g)
    do = torch.randn(m, c, generator=g)
    qg, kg, vg = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    o = K.sra_attention(qg, kg, vg, plan, heads, impl=impl)
    ref = sst_oracle.sra_core(q.numpy(), k.numpy(), v.numpy(), tok, off, heads)
    err = np.abs(o.detach().cpu().numpy() - ref).max()
    assert err < TOL, f'forward max abs err {err}'
    (o * do.to(DEV)).sum().backward()
    rdq, rdk, rdv = sst_oracle.sra_core_backward(q.numpy(), k.numpy(), v.numpy(), do.numpy(), tok, off, heads)
    for name_, got, want in (('dq', qg.grad, rdq), ('dk', kg.grad, rdk), ('dv', vg.grad, rdv)):
        e = np.abs(got.cpu().numpy() - want).max()
        assert e < TOL, f'{name_} max abs err {e}'


def test_sra_core_packed_qk_and_strided_inputs():
    from sst_amd import kernels as K
    from oracle import sst_oracle
    plan, tok, off, m = _plan_from_sizes([33, 70, 5, 120], 9)
    g = torch.Generator().manual_seed(1)
    qk = torch.randn(m, 256, generator=g)
    v = torch.randn(m, 128, generator=g)
    do = torch.randn(m, 128, generator=g)
    qkg, vg = qk.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    o = K.sra_attention_qk_v(qkg, vg, plan, 8)
    ref = sst_oracle.sra_core(qk[:, :128].numpy(), qk[:, 128:].numpy(), v.numpy(), tok, off, 8)
    assert np.abs(o.detach().cpu().numpy() - ref).max() < TOL
    (o * do.to(DEV)).sum().backward()
    rdq, rdk, rdv = sst_oracle.sra_core_backward(qk[:, :128].numpy(), qk[:, 128:].numpy(), v.numpy(), do.numpy(), tok,
                                                 off, 8)
    assert np.abs(qkg.grad[:, :128].cpu().numpy() - rdq).max() < TOL
    assert np.abs(qkg.grad[:, 128:].cpu().numpy() - rdk).max() < TOL
    assert np.abs(vg.grad.cpu().numpy() - rdv).max() < TOL


def test_sra_core_backward_full_size_vs_oracle():
    """M ~ 90k tokens (the bench workload's window-size mix): forward and the one-pass backward kernel against the
    float64 oracle on every 25th window (the oracle walks windows one by one), and against the two-launch and the
    generic kernels on every row."""
    from sst_amd import kernels as K
    from oracle import sst_oracle
    rng = np.random.default_rng(3)
    sizes = rng.integers(20, 101, size=1500).tolist()
    plan, tok, off, m = _plan_from_sizes(sizes, 5)
    g = torch.Generator().manual_seed(6)
    q, k, v, do = (torch.randn(m, 128, generator=g) for _ in range(4))
    grads = {}
    for impl in (0, 3, 1):
        qg, kg, vg = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        o = K.sra_attention(qg, kg, vg, plan, 8, impl=impl)
        (o * do.to(DEV)).sum().backward()
        grads[impl] = (o.detach().cpu().numpy(), qg.grad.cpu().numpy(), kg.grad.cpu().numpy(), vg.grad.cpu().numpy())
    for impl in (3, 1):
        for a, b in zip(grads[0], grads[impl]):
            assert np.abs(a - b).max() < 1e-4
    sel = np.arange(0, len(sizes), 25)
    sub_tok = np.concatenate([tok[off[w]:off[w + 1]] for w in sel])
    sub_off = np.concatenate([[0], np.cumsum([sizes[w] for w in sel])])
    rows = np.sort(sub_tok)
    ro = sst_oracle.sra_core(q.numpy(), k.numpy(), v.numpy(), sub_tok, sub_off, 8)
    rdq, rdk, rdv = sst_oracle.sra_core_backward(q.numpy(), k.numpy(), v.numpy(), do.numpy(), sub_tok, sub_off, 8)
    for name_, got, want in zip(('o', 'dq', 'dk', 'dv'), grads[0], (ro, rdq, rdk, rdv)):
        e = np.abs(got[rows] - want[rows]).max()
        assert e < 1e-4, f'{name_} max abs err {e} against the float64 oracle'


def test_sra_core_properties_full_size():
    """M ~ 90k tokens: rows of softmax sum to one (V = 1 -> O = 1), linearity in V, MFMA == generic."""
    from sst_amd import kernels as K
    rng = np.random.default_rng(0)
    sizes = rng.integers(20, 101, size=1500).tolist()
    plan, tok, off, m = _plan_from_sizes(sizes, 4)
    g = torch.Generator().manual_seed(2)
    q = torch.randn(m, 128, generator=g).to(DEV)
    k = torch.randn(m, 128, generator=g).to(DEV)
    v1 = torch.randn(m, 128, generator=g).to(DEV)
    v2 = torch.randn(m, 128, generator=g).to(DEV)
    ones = K.sra_attention(q, k, torch.ones_like(v1), plan, 8)
    assert float((ones - 1).abs().max()) < 1e-5
    o1 = K.sra_attention(q, k, v1, plan, 8)
    o2 = K.sra_attention(q, k, v2, plan, 8)
    o12 = K.sra_attention(q, k, v1 + 2 * v2, plan, 8)
    assert float((o12 - (o1 + 2 * o2)).abs().max()) < 1e-4
    og = K.sra_attention(q, k, v1, plan, 8, impl=1)
    assert float((og - o1).abs().max()) < 1e-4
    ol = K.sra_attention(q, k, v1, plan, 8, impl=2)
    assert float((ol - o1).abs().max()) < 1e-4


def test_sra_window_launch_order_changes_nothing_but_the_schedule(monkeypatch):
    """the plan's launch order (windows by ascending token count, dispatched from the end) is a permutation, and the
    forward output and the three gradients are bit for bit what the row order gives"""
    from sst_amd import kernels as K
    rng = np.random.default_rng(1)
    sizes = rng.integers(1, 101, size=1500).tolist()
    plan, tok, off, m = _plan_from_sizes(sizes, 4)
    order = plan.order
    assert order is not None and sorted(order.cpu().tolist()) == list(range(len(sizes)))
    assert (np.diff(np.asarray(sizes)[order.cpu().numpy()]) >= 0).all()
    g = torch.Generator().manual_seed(5)
    q, k, v, do = (torch.randn(m, 128, generator=g).to(DEV) for _ in range(4))
    outs = []
    for ordered in (True, False):
        monkeypatch.setattr(K, 'WINDOW_ORDER_MIN', 512 if ordered else 1 << 30)
        p = K.WindowPlan(plan.tok, plan.winoff, plan.n_windows, plan.n_tokens, plan.max_tokens)
        assert (p.order is not None) == ordered
        qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = K.sra_attention(qa, ka, va, p, 8)
        o.backward(do)
        outs.append((o.detach(), qa.grad, ka.grad, va.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _load_block(g, layer_cfg):
    import sst_amd
    d, h, ffn = int(g['cfg::d_model']), int(g['cfg::nhead']), int(g['cfg::ffn'])
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[d], nhead=[h], num_blocks=1, dim_feedforward=[ffn],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=True,
                                      layer_cfg=layer_cfg))
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}
    net.load_state_dict(sd, strict=True)   # same state_dict keys as the reference
    return net.to(DEV).train(), d


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('impl', [0, 1])
@pytest.mark.parametrize('tag,layer_cfg', [('std', dict()), ('cosine', dict(cosine=True, tau_min=0.01)),
                                           ('cosine_ns', dict(cosine=True, tau_min=0.01, non_shared_tau=True)),
                                           ('prenorm', dict(post_norm=False)),
                                           ('bn_cosine', dict(use_bn=True, cosine=True, tau_min=0.01))])
def test_sst_block_matches_reference_golden(tag, layer_cfg, impl, fused):
    import sst_amd
    g = load_golden(f'sst_block_{tag}.npz')
    net, d = _load_block(g, layer_cfg)
    net.set_impl(impl)
    net.set_fused(fused)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                    debug=True, mute=True)
    layer.eval()
    feats = torch.from_numpy(g['in::voxel_feats']).to(DEV).requires_grad_(True)
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)
    info = layer(feats, coors, 2)
    out = net(info)[0]['voxel_feats']
    err = np.abs(out.detach().cpu().numpy() - g['out::voxel_feats']).max()
    assert err < TOL, f'block output max abs err {err}'
    (out * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()
    e = np.abs(feats.grad.cpu().numpy() - g['out::grad_in']).max()
    assert e < TOL * 5, f'input gradient max abs err {e}'
    params = dict(net.named_parameters())
    for key in [k for k in g if k.startswith('grad::')]:
        got = params[key[6:]].grad.cpu().numpy()
        scale = max(1.0, float(np.abs(g[key]).max()))
        assert np.abs(got - g[key]).max() < TOL * 5 * scale, key


def test_sst_block_accepts_reference_style_dicts():
    """A voxel_info that only carries the reference's per-level dictionaries drives the same kernels."""
    import sst_amd
    g = load_golden('sst_block_std.npz')
    net, d = _load_block(g, dict())
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                    debug=True, mute=True)
    layer.eval()
    info = layer(torch.from_numpy(g['in::voxel_feats']).to(DEV), torch.from_numpy(g['in::voxel_coors']).to(DEV), 2)
    ref_style = {k: v for k, v in info.items() if not k.startswith('sra_plan') and not k.startswith('pos_embed')}
    with torch.no_grad():
        out = net(ref_style)[0]['voxel_feats']
    assert np.abs(out.cpu().numpy() - g['out::voxel_feats']).max() < TOL


def test_recover_bev_matches_loop_semantics():
    import sst_amd
    net = sst_amd.SSTv2(d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256], output_shape=[468, 468],
                        num_attached_conv=0, to_bev=True).to(DEV)
    g = torch.Generator().manual_seed(0)
    coors = torch.unique(torch.stack([torch.randint(0, 2, (3000,), generator=g), torch.zeros(3000, dtype=torch.long),
                                      torch.randint(0, 468, (3000,), generator=g),
                                      torch.randint(0, 468, (3000,), generator=g)], 1), dim=0).to(DEV)
    feat = torch.randn(coors.size(0), 128, device=DEV)
    bev = net.recover_bev(feat, coors, 2)
    assert bev.shape == (2, 128, 468, 468)
    ref = torch.zeros(2, 128, 468 * 468, device=DEV)
    for b in range(2):
        msk = coors[:, 0] == b
        ref[b][:, coors[msk, 2] * 468 + coors[msk, 3]] = feat[msk].t()
    assert torch.equal(bev, ref.view(2, 128, 468, 468))


def test_sst_v1_matches_reference_golden():
    """First-generation twins: SSTInputLayer + SSTv1 vs the reference's own classes (eval mode, no drop)."""
    import sst_amd
    from conftest import PC_RANGE, VOXEL_SIZE
    g = load_golden('sst_v1.npz')
    layer = sst_amd.build_middle_encoder(dict(
        type='SSTInputLayer', drop_info=(DROP_TRAIN, DROP_TEST), shifts_list=[(0, 0), (6, 6)], window_shape=(12, 12),
        point_cloud_range=PC_RANGE, voxel_size=VOXEL_SIZE, shuffle_voxels=False, debug=True))
    layer.eval()
    net = sst_amd.build_backbone(dict(
        type='SSTv1', d_model=[64, 64], nhead=[4, 4], num_blocks=2, dim_feedforward=[128, 128],
        output_shape=[468, 468], num_attached_conv=0, debug=True, drop_info=(DROP_TRAIN, DROP_TEST),
        pos_temperature=10000, normalize_pos=False, window_shape=(12, 12)))
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}
    net.load_state_dict(sd, strict=True)
    net.to(DEV).eval()
    feats = torch.from_numpy(g['in::voxel_feats']).to(DEV)
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)
    with torch.no_grad():
        vf, ind_list, info = layer(feats, coors, 2)     # third positional argument accepted
        bev = net((vf, ind_list, info))[0]
    np.testing.assert_array_equal(info['voxel_keep_inds'].cpu().numpy(), g['out::voxel_keep_inds'])
    np.testing.assert_array_equal(info['coors'].cpu().numpy(), g['out::coors'])
    for s in range(2):
        for k in (f'batch_win_inds_shift{s}', f'coors_in_win_shift{s}', f'voxel_drop_level_shift{s}'):
            np.testing.assert_array_equal(info[k].cpu().numpy(), g['out::' + k])
    assert bev.shape == (2, 64, 468, 468)
    c = info['coors']
    got = bev[c[:, 0], :, c[:, 2], c[:, 3]].cpu().numpy()
    assert np.abs(got - g['out::bev_at_voxels']).max() < TOL
    assert abs(float(bev.abs().sum()) - float(g['out::bev_abs_sum'])) < 1e-2 * max(1.0, float(g['out::bev_abs_sum']))
    # the reference's own index dictionaries drive the same kernels
    with torch.no_grad():
        bev2 = net((vf, ind_list, {k: v for k, v in info.items() if not k.startswith('sra_plan')}))[0]
    assert float((bev2 - bev).abs().max()) < 1e-5


@pytest.mark.parametrize('tag,shortcut', [('plain', False), ('shortcut', True)])
def test_sstv2_bev_and_attached_convs_match_reference_golden(tag, shortcut):
    """the output side of SSTv2 (a14): recover_bev + two attached dilated convolutions with naiveSyncBN2d + ReLU
    (sst_v2.py:86-92, 139-197), training mode, against the reference's own SSTv2 (tests/golden/sst_bev_*.npz):
    dense canvas, input gradient, convolution weight gradients."""
    import sst_amd
    g = load_golden(f'sst_bev_{tag}.npz')
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[32], nhead=[2], num_blocks=1, dim_feedforward=[64],
                                      output_shape=[48, 48], num_attached_conv=2, conv_in_channel=32,
                                      conv_out_channel=32, debug=True, to_bev=True, conv_shortcut=shortcut))
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).train()
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (48, 48, 1), shuffle_voxels=False, debug=True,
                                    mute=True)
    layer.eval()
    feats = torch.from_numpy(g['in::voxel_feats']).to(DEV).requires_grad_(True)
    info = layer(feats, torch.from_numpy(g['in::voxel_coors']).to(DEV), 2)
    bev = net(info)[0]
    assert tuple(bev.shape) == (2, 32, 48, 48)
    err = np.abs(bev.detach().cpu().numpy() - g['out::bev']).max()
    assert err < TOL * max(1.0, float(np.abs(g['out::bev']).max())), err
    (bev * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()
    e = np.abs(feats.grad.cpu().numpy() - g['out::grad_in']).max()
    assert e < TOL * 5 * max(1.0, float(np.abs(g['out::grad_in']).max())), e
    params = dict(net.named_parameters())
    for key in [k for k in g if k.startswith('grad::')]:
        got = params[key[6:]].grad.cpu().numpy()
        assert np.abs(got - g[key]).max() < TOL * 5 * max(1.0, float(np.abs(g[key]).max())), key


@pytest.mark.parametrize('heads', [4, 2])
def test_composed_attention_other_head_dims_vs_oracle(heads):
    """head_dim 32 / 64 (nn.MultiheadAttention accepts any divisor; no SST config uses them): composed path, fwd + bwd"""
    from sst_amd.sra_composed import sra_attention_composed
    from sst_amd import kernels as K
    from oracle import sst_oracle
    rng = np.random.default_rng(heads)
    sizes = [1, 5, 16, 30, 47, 60, 3, 100]
    m = sum(sizes)
    tok = rng.permutation(m).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    plan = K.WindowPlan(torch.from_numpy(tok).to(DEV), torch.from_numpy(off).to(DEV), len(sizes), m, max(sizes))
    g = torch.Generator().manual_seed(heads)
    q, k, v, do = (torch.randn(m, 128, generator=g) for _ in range(4))
    qg, kg, vg = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    scale = 1.0 / (128 // heads) ** 0.5
    o = sra_attention_composed(qg, kg, vg, plan, heads, scale)
    ref = sst_oracle.sra_core(q.numpy(), k.numpy(), v.numpy(), tok, off, heads)
    assert np.abs(o.detach().cpu().numpy() - ref).max() < 1e-4
    (o * do.to(DEV)).sum().backward()
    for got, want in zip((qg.grad, kg.grad, vg.grad), sst_oracle.sra_core_backward(q.numpy(), k.numpy(), v.numpy(),
                                                                                  do.numpy(), tok, off, heads)):
        assert np.abs(got.cpu().numpy() - want).max() < 1e-3


def test_encoder_layer_attention_dropout_and_head_dim_32():
    """attention-weight dropout > 0: identity in eval mode, active in training; a head_dim-32 layer runs end to end"""
    import sst_amd
    from sst_amd import kernels as K
    rng = np.random.default_rng(3)
    sizes = [20, 33, 60, 7]
    m = sum(sizes)
    plan = K.WindowPlan(torch.from_numpy(rng.permutation(m).astype(np.int32)).to(DEV),
                        torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(DEV), len(sizes), m,
                        max(sizes))
    torch.manual_seed(0)
    plain = sst_amd.EncoderLayer(128, 8, 256, dropout=0.0).to(DEV)
    drop = sst_amd.EncoderLayer(128, 8, 256, dropout=0.2).to(DEV)
    drop.load_state_dict(plain.state_dict())
    x = torch.randn(m, 128, device=DEV)
    pos = torch.randn(m, 128, device=DEV)
    plain.eval(), drop.eval()
    with torch.no_grad():
        assert torch.allclose(plain(x, pos, plan), drop(x, pos, plan), atol=1e-5)
    drop.train()
    xa = x.clone().requires_grad_(True)
    out = drop(xa, pos, plan)
    out.square().sum().backward()
    assert torch.isfinite(out).all() and torch.isfinite(xa.grad).all()
    with torch.no_grad():
        assert float((out - plain(x, pos, plan)).abs().max()) > 1e-3          # the dropout did something
    wide = sst_amd.EncoderLayer(128, 4, 256, dropout=0.0).to(DEV).train()     # head_dim 32
    xb = x.clone().requires_grad_(True)
    y = wide(xb, pos, plan)
    y.square().sum().backward()
    assert y.shape == x.shape and torch.isfinite(xb.grad).all()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fp32', 'f32x6'])
def test_checkpoint_blocks_recompute_inside_the_fused_chain(mode, monkeypatch):
    """configs with checkpoint_blocks (sst_v2.py:131-133: torch.utils.checkpoint per shift block when training) stay on the
    chain of fused layer nodes - the listed blocks are recomputed in the backward pass - instead of falling back to the
    per-layer path: same outputs and gradients as without checkpointing, bit for bit (deterministic kernels)"""
    import sst_amd
    from conftest import DROP_TEST, DROP_TRAIN
    from sst_amd import sst_basic_block as SB
    g = torch.Generator().manual_seed(4)
    side = 60
    cells = torch.randperm(2 * side * side, generator=g)[:2500].sort()[0]
    coors = torch.stack([cells // (side * side), torch.zeros_like(cells), cells // side % side + 50, cells % side + 50], 1).to(DEV)
    feats0 = torch.randn(2500, 128, generator=g).to(DEV)
    up = torch.randn(2500, 128, generator=g).to(DEV)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True, reference_outputs=False).eval()
    results = {}
    for ckpt in ([], [0, 2]):
        torch.manual_seed(9)
        net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128] * 3, nhead=[8] * 3, num_blocks=3, dim_feedforward=[256] * 3,
                                          output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False,
                                          checkpoint_blocks=ckpt)).to(DEV).train()
        net.set_precision(mode)
        try:
            if ckpt:   # the per-layer path must not be taken
                monkeypatch.setattr(SB.BasicShiftBlockV2, 'forward', lambda *a, **k: (_ for _ in ()).throw(AssertionError('per-layer path')))
            feats = feats0.clone().requires_grad_(True)
            out = net(layer(feats, coors, 2))[0]['voxel_feats']
            (out * up[:out.size(0)]).sum().backward()
        finally:
            net.set_precision('fp32')
        results[bool(ckpt)] = (out.detach().clone(), feats.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
    (o0, g0, p0), (o1, g1, p1) = results[False], results[True]
    assert torch.equal(o0, o1) and torch.equal(g0, g1)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n


@pytest.mark.parametrize('cap', [60, 80, 100, 144])
def test_window_ordered_rows_need_no_token_list(cap):
    """feature rows already in window order (the frame plan's unshifted partition): the kernels take d_tok = NULL and return
    the same bits as with the explicit list arange(M), forward and backward"""
    from sst_amd import kernels as K
    torch.manual_seed(cap)
    sizes = torch.randint(1, cap + 1, (300,))
    winoff = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(torch.int32).to(DEV)
    m = int(sizes.sum())
    tok = torch.arange(m, dtype=torch.int32, device=DEV)
    qkv = torch.randn(m, 384, device=DEV)
    do = torch.randn(m, 128, device=DEV)
    q, k, v = qkv[:, :128], qkv[:, 128:256], qkv[:, 256:]
    outs = []
    for flag in (False, True):
        plan = K.WindowPlan(tok, winoff, 300, m, cap, rows_in_window_order=flag)
        assert (plan.tok_ptr(0) is None) == flag
        o, lse = K._sra_fwd(q, k, v, plan, 8, 0.25, 0)
        dqkv = torch.empty_like(qkv)
        K._sra_bwd(q, k, v, o, lse, do, plan, 8, 0.25, 0, dqkv[:, :128], dqkv[:, 128:256], dqkv[:, 256:])
        outs.append((o, lse, dqkv))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('fused', [True, False])
def test_d_model_192_twelve_heads_against_float64(fused):
    """configs/sst/sst_waymoD5_1x_3class_12heads.py:56-80 (d_model 192, 12 heads, feed-forward 384): the attention kernels take any
    multiple of four heads of width 16; the projections of that width run on the library / LDS-resident fp32 kernels.  Forward of
    one shift block against the float64 restatement (oracle/sst_oracle.encoder_layer), and its gradients finite."""
    import sst_amd
    from oracle import sst_oracle
    torch.manual_seed(4)
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[192], nhead=[12], num_blocks=1, dim_feedforward=[384],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False)).to(DEV).train()
    net.set_fused(fused)
    g = torch.Generator().manual_seed(5)
    side, n = 80, 2600
    cells = torch.randperm(side * side, generator=g)[:n].sort()[0]
    coors = torch.stack([torch.zeros_like(cells), torch.zeros_like(cells), cells // side + 30, cells % side + 30], 1).to(DEV)
    feats = torch.randn(n, 192, generator=g).to(DEV).requires_grad_(True)
    layer = sst_amd.SSTInputLayerV2((DROP_TEST, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True).eval()
    info = layer(feats, coors, 1)
    out = net(info)[0]['voxel_feats']
    x = info['voxel_feats'].detach().double().cpu().numpy()
    for li, enc in enumerate(net.block_list[0].encoder_list):
        plan = info[f'sra_plan_shift{li}']
        pos = info[f'pos_embed_shift{li}'].double().cpu().numpy()
        params = {k: v.detach().double().cpu().numpy() for k, v in enc.state_dict().items()}
        x = sst_oracle.encoder_layer(x, pos, plan.tok.cpu().numpy(), plan.winoff[:plan.n_windows + 1].cpu().numpy(), params, 12)
    err = float(np.abs(out.detach().double().cpu().numpy() - x).max())
    assert err < 2e-5, err
    out.square().sum().backward()
    assert torch.isfinite(feats.grad).all() and float(feats.grad.abs().max()) > 0
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def _run_split_probe(split):
    """forward of the attention core on 12 windows of 49..100 tokens (a SMALL launch: the query-tile split is on by default) in a
    process of its own with SST_SRA_SPLIT=<split> (the switch is read once per process) -> output tensors on the host"""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from sst_amd import kernels as K
sizes = np.array([100, 97, 81, 80, 65, 64, 49, 52, 33, 16, 7, 100], dtype=np.int64)
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
m = int(off[-1])
g = torch.Generator().manual_seed(11)
qkv = torch.randn(m, 384, generator=g).cuda()
perm = torch.randperm(m, generator=g).int().cuda()
for tag, plan in (('rows', K.WindowPlan(torch.arange(m, dtype=torch.int32).cuda(), torch.from_numpy(off).cuda(), len(sizes), m, 100, rows_in_window_order=True)),
                  ('list', K.WindowPlan(perm, torch.from_numpy(off).cuda(), len(sizes), m, 100))):
    o, lse = K._sra_fwd(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], plan, 8, 0.25, 0)
    hs = torch.linspace(2.0, 6.0, 8).cuda()
    oc, lsec = K._sra_cos_fwd(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], plan, 8, hs)
    do = torch.randn(m, 128, generator=torch.Generator().manual_seed(12)).cuda()
    out = dict(o=o.cpu().numpy(), lse=lse.cpu().numpy(), oc=oc.cpu().numpy(), lsec=lsec.cpu().numpy())
    for rep in range(2):      # twice: the hand-over of the partial dK / dV tiles must not depend on which wave arrives first
        d = torch.empty_like(qkv)
        K._sra_bwd(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], o, lse, do, plan, 8, 0.25, 0, d[:, :128], d[:, 128:256], d[:, 256:])
        dc = torch.empty_like(qkv)
        K._sra_cos_bwd(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], oc, lsec, do, plan, 8, hs, dc[:, :128], dc[:, 128:256], dc[:, 256:])
        out['d' + str(rep)], out['dc' + str(rep)] = d.cpu().numpy(), dc.cpu().numpy()
    np.savez(sys.argv[1] + tag + '.npz', **out)
''' % (ROOT,)
    d = tempfile.mkdtemp()
    env = dict(os.environ, SST_SRA_SPLIT=str(split))
    subprocess.run([sys.executable, '-c', code, d + '/'], check=True, env=env, timeout=300)
    return {t: dict(np.load(f'{d}/{t}.npz')) for t in ('rows', 'list')}


def test_query_tile_split_of_small_launches():
    """sra_fwd_wave_k deals windows of >= 4 tiles out over two (or four) workgroups when the launch is small: every query row is
    still computed by one wave from the same K / V fragments in the same order - the outputs must not differ in a single bit from
    the unsplit launch (standard and cosine attention, rows in window order and through a token list)"""
    base = _run_split_probe(1)
    for parts in (2, 4):
        got = _run_split_probe(parts)
        for t in base:
            for k in ('o', 'lse', 'oc', 'lsec'):
                assert np.array_equal(base[t][k], got[t][k]), (parts, t, k)
            # backward (two parts whatever SST_SRA_SPLIT > 1 says): dQ rows belong to one wave - bit-identical; dK / dV are the
            # sum of two partial sums over the query tiles instead of one running sum - equal to fp32 rounding, and reproducible
            for k in ('d', 'dc'):
                assert np.array_equal(got[t][k + '0'], got[t][k + '1']), (parts, t, k, 'run-to-run')
                assert np.array_equal(base[t][k + '0'][:, :128], got[t][k + '0'][:, :128]), (parts, t, k, 'dQ')
                scale = np.abs(base[t][k + '0']).max()
                assert np.abs(base[t][k + '0'] - got[t][k + '0']).max() <= 2e-6 * scale, (parts, t, k)
