"""GPU: the reduced-precision (bf16) mode - attention core, LayerNorm kernels, encoder layers.

Pins: the float64 oracle on bf16-rounded inputs for the kernels; for the block, tensors produced by the reference's own
SSTv2 under torch.autocast(bfloat16) on CPU (tests/golden/sst_block_bf16.npz, made by tests/golden/make_golden.py).
Tolerances are stated at bf16 resolution: a bf16 value carries 8 significant bits, so a stored activation of magnitude
~4 is itself only known to +-0.016; the reference's autocast run differs from its own fp32 run by 2.0e-2 (max abs) on
this block."""
import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF = torch.bfloat16


def _plan_from_sizes(sizes, seed):
    from sst_amd import kernels as K
    rng = np.random.default_rng(seed)
    m = int(sum(sizes))
    tok = rng.permutation(m).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    plan = K.WindowPlan(torch.from_numpy(tok).to(DEV), torch.from_numpy(off).to(DEV), len(sizes), m, max(sizes))
    return plan, tok, off, m


@pytest.mark.parametrize('heads', [8, 12])
@pytest.mark.parametrize('sizes', [[1, 2, 3, 15, 16, 17], [30, 31, 33, 47, 48, 49, 60, 63, 64, 65, 79, 80, 81],
                                   [96, 100, 111, 112, 113, 128, 143, 144], [1, 144, 7, 100, 64, 30, 16, 59, 81, 12, 5]])
def test_sra_core_bf16_vs_oracle(sizes, heads):
    from sst_amd import bf16
    from oracle import sst_oracle
    plan, tok, off, m = _plan_from_sizes(sizes, len(sizes) + heads)
    c = heads * 16
    g = torch.Generator().manual_seed(m + heads)
    q, k, v, do = ((torch.randn(m, c, generator=g) * s).to(BF) for s in (1.5, 1.5, 1.0, 1.0))
    qg, kg, vg = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    o = bf16.sra_attention(qg, kg, vg, plan, heads)
    assert o.dtype == BF
    f = lambda t: t.float().numpy()
    ref = sst_oracle.sra_core(f(q), f(k), f(v), tok, off, heads)
    err = np.abs(o.detach().float().cpu().numpy() - ref).max()
    assert err < 2e-2, f'forward max abs err {err}'          # |o| <= max |v| ~ 4: bf16 output rounding alone is 1.6e-2
    (o.float() * do.to(DEV).float()).sum().backward()
    rdq, rdk, rdv = sst_oracle.sra_core_backward(f(q), f(k), f(v), f(do), tok, off, heads)
    for name, got, want in (('dq', qg.grad, rdq), ('dk', kg.grad, rdk), ('dv', vg.grad, rdv)):
        e = np.abs(got.float().cpu().numpy() - want).max()
        assert e < 3e-2 * max(1.0, np.abs(want).max()), f'{name} max abs err {e} (scale {np.abs(want).max()})'


def test_sra_core_bf16_full_size_against_fp32_kernels():
    """M ~ 90k tokens: bf16 kernels against the fp32 kernels on the same (bf16-representable) inputs."""
    from sst_amd import bf16, kernels as K
    rng = np.random.default_rng(3)
    sizes = rng.integers(20, 101, size=1500).tolist()
    plan, tok, off, m = _plan_from_sizes(sizes, 5)
    g = torch.Generator().manual_seed(6)
    q, k, v, do = (torch.randn(m, 128, generator=g).to(BF) for _ in range(4))
    qa, ka, va = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    oa = bf16.sra_attention(qa, ka, va, plan, 8)
    (oa.float() * do.to(DEV).float()).sum().backward()
    qb, kb, vb = (t.float().to(DEV).requires_grad_(True) for t in (q, k, v))
    ob = K.sra_attention(qb, kb, vb, plan, 8)
    (ob * do.to(DEV).float()).sum().backward()
    assert float((oa.detach().float() - ob.detach()).abs().max()) < 2e-2
    for a, b in ((qa, qb), (ka, kb), (va, vb)):
        scale = max(1.0, float(b.grad.abs().max()))
        assert float((a.grad.float() - b.grad).abs().max()) < 3e-2 * scale
        assert float((a.grad.float() - b.grad).abs().mean()) < 3e-3 * scale


@pytest.mark.parametrize('m,c', [(1, 128), (77, 128), (5000, 192), (90107, 128)])
def test_layernorm_bf16_kernels(m, c):
    from sst_amd import bf16
    g = torch.Generator().manual_seed(m + c)
    x, r, dy, dy2 = (torch.randn(m, c, generator=g).to(BF).to(DEV) for _ in range(4))
    w = (1 + 0.1 * torch.randn(c, generator=g)).to(DEV)
    b = (0.1 * torch.randn(c, generator=g)).to(DEV)
    table = torch.randn(144, c, generator=g).to(DEV)
    idx = torch.randint(0, 144, (m,), generator=g).to(torch.int32).to(DEV)
    y, s, stats, yp = bf16.add_ln_fwd(x, r, w, b, 1e-5, pos=(table, idx))
    xs = (x.float() + r.float()).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xs, (c,), w, b, 1e-5)
    assert float((y.float() - ref.detach()).abs().max()) < 3e-2 and float((y.float() - ref.detach()).abs().mean()) < 3e-3
    assert float((yp.float() - (ref.detach() + table[idx.long()])).abs().max()) < 5e-2
    assert float((s.float() - xs.detach()).abs().max()) < 3e-2
    dx, dw, db = bf16.add_ln_bwd(dy, dy2, s, stats, w)
    gsum = dy.float() + dy2.float()
    # reference gradient on the SAME rounded sum the kernel saw
    xs2 = s.float().requires_grad_(True)
    w2 = w.clone().requires_grad_(True)
    b2 = b.clone().requires_grad_(True)
    (torch.nn.functional.layer_norm(xs2, (c,), w2, b2, 1e-5) * gsum).sum().backward()
    assert float((dx.float() - xs2.grad).abs().max()) < 3e-2 * max(1.0, float(xs2.grad.abs().max()))
    assert float((dw - w2.grad).abs().max()) < 2e-3 * max(1.0, float(w2.grad.abs().max()))
    assert float((db - b2.grad).abs().max()) < 2e-3 * max(1.0, float(b2.grad.abs().max()))
    out = bf16.cast_add_pos(x.float(), (table, idx))
    assert float((out.float() - (x.float() + table[idx.long()])).abs().max()) < 3e-2


@pytest.mark.parametrize('m', [1, 77, 5000, 90107])
@pytest.mark.parametrize('k,n', [(128, 128), (128, 256), (256, 128)])
def test_tall_linear_bf16(m, k, n):
    """y = epilogue(x w^T + b) against fp32 torch on the same bf16-rounded operands; every epilogue"""
    from sst_amd import bf16
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).to(BF).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(BF).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    aux = torch.randn(m, n, generator=g).to(BF).to(DEV)
    ref = x.float() @ w.float().t() + b
    tol = lambda r: 1.6e-2 * max(1.0, float(r.abs().max()))          # one bf16 rounding of the output
    y = bf16.tall_linear(x, w, b)
    assert y.dtype == BF and float((y.float() - ref).abs().max()) < tol(ref)
    y0 = bf16.tall_linear(x, w)                                       # no bias
    assert float((y0.float() - (ref - b)).abs().max()) < tol(ref)
    for epi, fn in ((bf16.EPI_GELU, torch.nn.functional.gelu), (bf16.EPI_RELU, torch.relu)):
        y, pre = bf16.tall_linear(x, w, b, epi, want_pre=True)
        assert float((pre.float() - ref).abs().max()) < tol(ref)
        assert float((y.float() - fn(pre.float())).abs().max()) < 1.6e-2 * max(1.0, float(ref.abs().max()))
    xg = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(xg).sum().backward()
    y = bf16.tall_linear(x, w, None, bf16.EPI_MUL_GELU_GRAD, aux_in=aux)
    want = (ref - b) * xg.grad
    assert float((y.float() - want).abs().max()) < tol(want)
    y = bf16.tall_linear(x, w, None, bf16.EPI_MUL_RELU_GRAD, aux_in=aux)
    want = (ref - b) * (aux.float() > 0)
    assert float((y.float() - want).abs().max()) < tol(want)
    y = bf16.tall_linear(x, w, None, bf16.EPI_ADD, aux_in=aux)
    want = (ref - b) + aux.float()
    assert float((y.float() - want).abs().max()) < tol(want)


@pytest.mark.parametrize('m', [1, 77, 5000, 90107])
@pytest.mark.parametrize('k', [128, 256])
def test_linear_add_layernorm_bf16(m, k):
    """projection + residual + LayerNorm (+ positional second output) in one kernel against fp32 torch"""
    from sst_amd import bf16
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).to(BF).to(DEV)
    res = torch.randn(m, 128, generator=g).to(BF).to(DEV)
    w = (torch.randn(128, k, generator=g) / k ** 0.5).to(BF).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    lw = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV)
    lb = (0.1 * torch.randn(128, generator=g)).to(DEV)
    table = torch.randn(144, 128, generator=g).to(DEV)
    idx = torch.randint(0, 144, (m,), generator=g).to(torch.int32).to(DEV)
    y, s, stats, yp = bf16.linear_add_ln(x, w, b, res, lw, lb, 1e-5, pos=(table, idx))
    ssum = x.float() @ w.float().t() + b + res.float()
    ref = torch.nn.functional.layer_norm(ssum, (128,), lw, lb, 1e-5)
    assert float((s.float() - ssum).abs().max()) < 1.6e-2 * max(1.0, float(ssum.abs().max()))
    assert float((y.float() - ref).abs().max()) < 3e-2 and float((y.float() - ref).abs().mean()) < 3e-3
    assert float((yp.float() - (ref + table[idx.long()])).abs().max()) < 5e-2
    mean, var = ssum.mean(1), ssum.var(1, unbiased=False)
    assert float((stats[:, 0] - mean).abs().max()) < 1e-3
    assert float((stats[:, 1] - torch.rsqrt(var + 1e-5)).abs().max()) < 1e-3 * float(torch.rsqrt(var + 1e-5).max())
    y2, s2, _, none = bf16.linear_add_ln(x, w, b, res, lw, lb, 1e-5, save_sum=False)
    assert s2 is None and none is None and torch.equal(y2, y)


@pytest.mark.parametrize('m', [1, 31, 4097, 90107])
def test_wgrad_group_bf16(m):
    """the five parameter-gradient products of one encoder layer in one launch, against fp64 on the same operands"""
    from sst_amd import bf16
    g = torch.Generator().manual_seed(m)
    mk = lambda c: torch.randn(m, c, generator=g).to(BF).to(DEV)
    dqkv, xp, x, ds1, o, dpre, y1, h, ds2 = mk(384), mk(128), mk(128), mk(128), mk(128), mk(256), mk(128), mk(256), mk(128)
    f32 = dict(dtype=torch.float32, device=DEV)
    dw_in, db_in = torch.full((384, 128), 7.0, **f32), torch.full((384,), 7.0, **f32)
    dwo, dbo = torch.empty((128, 128), **f32), torch.empty(128, **f32)
    dw1, db1 = torch.empty((256, 128), **f32), torch.empty(256, **f32)
    dw2, db2 = torch.empty((128, 256), **f32), torch.empty(128, **f32)
    bf16.wgrad_group([(dqkv[:, :256], xp, dw_in[:256], db_in[:256], 1, 0), (dqkv[:, 256:], x, dw_in[256:], db_in[256:], 1, 0),
                      (ds1, o, dwo, dbo, 1, 0), (dpre, y1, dw1, db1, 1, 0), (h, ds2, dw2, db2, 2, 1)])
    d = lambda t: t.double()
    want = [(d(dqkv[:, :256]).t() @ d(xp), dw_in[:256]), (d(dqkv[:, 256:]).t() @ d(x), dw_in[256:]), (d(ds1).t() @ d(o), dwo),
            (d(dpre).t() @ d(y1), dw1), (d(ds2).t() @ d(h), dw2), (d(dqkv).sum(0), db_in), (d(ds1).sum(0), dbo),
            (d(dpre).sum(0), db1), (d(ds2).sum(0), db2)]
    for ref, got in want:
        # fp32 accumulation of m exact bf16 x bf16 products: error ~ sqrt(m) * 2^-24 * |term|
        assert float((got.double() - ref).abs().max()) < 2e-5 * max(1.0, m ** 0.5) * 4


def test_sst_block_bf16_matches_reference_autocast_golden():
    import sst_amd
    g = load_golden('sst_block_bf16.npz')
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=True))
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}, strict=True)
    net = net.to(DEV).train().set_precision('bf16')
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                    debug=True, mute=True, reference_outputs=False)
    layer.eval()
    feats = torch.from_numpy(g['in::voxel_feats']).to(DEV).requires_grad_(True)
    info = layer(feats, torch.from_numpy(g['in::voxel_coors']).to(DEV), 2)
    out = net(info)[0]['voxel_feats']
    assert out.dtype == torch.float32
    d = np.abs(out.detach().cpu().numpy() - g['out::voxel_feats'])
    ref_gap = np.abs(g['out::voxel_feats'] - g['out::voxel_feats_fp32']).max()   # autocast vs fp32 in the reference
    assert d.max() < 6e-2 and d.mean() < 8e-3, (d.max(), d.mean(), ref_gap)
    d32 = np.abs(out.detach().cpu().numpy() - g['out::voxel_feats_fp32'])
    assert d32.max() < 6e-2 and d32.mean() < 8e-3
    (out * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()
    scale = max(1.0, float(np.abs(g['out::grad_in']).max()))
    e = np.abs(feats.grad.cpu().numpy() - g['out::grad_in'])
    assert e.max() < 8e-2 * scale and e.mean() < 8e-3 * scale, (e.max(), e.mean())
    params = dict(net.named_parameters())
    for key in [k for k in g if k.startswith('grad::')]:
        got = params[key[6:]].grad
        assert got.dtype == torch.float32
        sc = max(1.0, float(np.abs(g[key]).max()))
        assert np.abs(got.cpu().numpy() - g[key]).max() < 5e-2 * sc, key


def test_bf16_mode_leaves_unsupported_layers_in_fp32():
    """batch-norm layers (and widths other than 128 / 256) are not covered by the bf16 kernels: set_precision('bf16') must then
    give the fp32 result bit for bit"""
    import sst_amd
    g = load_golden('sst_block_bn_cosine.npz')
    d, h, ffn = int(g['cfg::d_model']), int(g['cfg::nhead']), int(g['cfg::ffn'])
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[d], nhead=[h], num_blocks=1, dim_feedforward=[ffn],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=True,
                                      layer_cfg=dict(use_bn=True, cosine=True, tau_min=0.01)))
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}, strict=True)
    net = net.to(DEV).train()
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                    debug=True, mute=True, reference_outputs=False)
    layer.eval()
    feats = torch.from_numpy(g['in::voxel_feats']).to(DEV)
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)
    with torch.no_grad():
        a = net(layer(feats, coors, 2))[0]['voxel_feats']
        b = net.set_precision('bf16')(layer(feats, coors, 2))[0]['voxel_feats']
    assert torch.equal(a, b)


def test_bf16_shadow_weights_follow_every_kind_of_parameter_update():
    """`.data` writes (mmcv's EMAHook swap, a master-to-model copy) do not move the version counter of a
    parameter.  The bf16 copies are re-made by one grouped launch at every forward, so the bf16 forward must follow the
    fp32 forward after optimizer.step(), after p.data.copy_() and after p.data.mul_()."""
    import sst_amd
    from sst_amd import bf16
    g = load_golden('sst_block_bf16.npz')
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=True))
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}, strict=True)
    net = net.to(DEV).train()
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                    debug=True, mute=True, reference_outputs=False)
    layer.eval()
    feats = torch.from_numpy(g['in::voxel_feats']).to(DEV)
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)

    def both():
        with torch.no_grad():
            a = net.set_precision('fp32')(layer(feats, coors, 2))[0]['voxel_feats']
            b = net.set_precision('bf16')(layer(feats, coors, 2))[0]['voxel_feats']
        return a, b

    def gap(a, b):      # bf16 resolution is relative: measure against the size of the outputs
        return float((a - b).abs().max()) / max(1.0, float(a.abs().max()))

    a0, b0 = both()
    assert gap(a0, b0) < 6e-2
    enc = net.block_list[0].encoder_list[0]
    # 1. an optimizer step (moves the version counter)
    opt = torch.optim.SGD(net.parameters(), lr=1.0)
    out = net.set_precision('bf16')(layer(feats.clone().requires_grad_(True), coors, 2))[0]['voxel_feats']
    out.square().mean().backward()
    gen0 = torch.Generator().manual_seed(3)
    for p in net.parameters():      # a step of visible size whatever the loss scale
        p.grad = (torch.randn(p.shape, generator=gen0) * 0.5).to(DEV) * (p.detach().std() if p.dim() > 1 else 0.1)
    opt.step()
    a1, b1 = both()
    assert gap(a1, a0) > 0.1, 'the step must change the network visibly'
    assert gap(a1, b1) < 6e-2
    # 2. p.data.copy_() and p.data.mul_(): no version bump
    gen = torch.Generator().manual_seed(1)
    v0 = enc.linear1.weight._version
    enc.linear1.weight.data.copy_(torch.randn(enc.linear1.weight.shape, generator=gen).to(DEV) * 0.3)
    enc.win_attn.self_attn.in_proj_weight.data.mul_(-1.5)
    enc.win_attn.self_attn.out_proj.weight.data.mul_(2.0)
    assert enc.linear1.weight._version == v0
    a2, b2 = both()
    assert gap(a2, a1) > 0.1
    assert gap(a2, b2) < 8e-2
    # the copies themselves: bit-equal to torch's round-to-nearest-even cast, both orientations, row ranges
    w = enc.win_attn.self_attn.in_proj_weight
    bf16.refresh_shadows([(w, (0, 256), False), (w, (256, 384), True), (enc.linear2.weight, None, True)])
    assert torch.equal(bf16.shadow(w, (0, 256)), w.detach()[:256].to(BF))
    assert torch.equal(bf16.shadow(w, (256, 384), transposed=True), w.detach()[256:].t().to(BF).contiguous())
    assert torch.equal(bf16.shadow(enc.linear2.weight, transposed=True), enc.linear2.weight.detach().t().to(BF).contiguous())
    odd = torch.nn.Parameter(torch.randn(45, 70, device=DEV))
    assert torch.equal(bf16.shadow(odd, (3, 40), transposed=True), odd.detach()[3:40].t().to(BF).contiguous())
    assert torch.equal(bf16.shadow(odd), odd.detach().to(BF))


@pytest.mark.parametrize('n_points,blocks,act', [(20000, 2, 'gelu'), (116000, 1, 'gelu'), (3000, 1, 'relu')])
def test_bf16_layer_executor_equals_the_python_sequence(n_points, blocks, act):
    """the reduced-precision encoder layer as ONE library call per direction (csrc/layer_exec.hip,
    sst_encoder_layer_{fwd,bwd}_bf16) against the same launch sequence issued from Python (sst_amd/bf16.py EncoderLayerBF16Fn):
    same kernels in the same order, so outputs and every gradient agree BIT FOR BIT through the whole pipeline"""
    import bench
    from sst_amd import bf16 as B
    torch.manual_seed(0)
    model = bench.Pipeline(blocks).to(DEV).train()
    model.backbone.set_precision('bf16')
    if act == 'relu':
        for blk in model.backbone.block_list:
            for enc in blk.encoder_list:
                enc.act_name, enc.activation = 'relu', torch.nn.functional.relu
    frames = [bench.make_cloud(n_points, 5, DEV)]
    calls = []
    orig = B._exec_fwd

    def run(exec_on, grad=True):
        B._LAYER_EXEC = 1 if exec_on else 0
        try:
            torch.manual_seed(11)                    # voxel shuffle / drop
            for p in model.parameters():
                p.grad = None
            if not grad:
                with torch.no_grad():
                    return model(frames).clone(), None
            out = model(frames)
            gen = torch.Generator(device=DEV).manual_seed(3)
            out.backward(torch.randn(out.shape, device=DEV, generator=gen))
            return out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            B._LAYER_EXEC = 1

    B._exec_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        out_e, grads_e = run(True)
    finally:
        B._exec_fwd = orig
    assert len(calls) == 2 * blocks, 'the executor did not take the layers'
    out_p, grads_p = run(False)
    assert torch.equal(out_e, out_p)
    assert grads_e.keys() == grads_p.keys() and len(grads_e) > 10
    for n in grads_e:
        assert torch.equal(grads_e[n], grads_p[n]), n
    out_ne, _ = run(True, grad=False)
    assert torch.equal(out_ne, out_e)


@pytest.mark.parametrize('layer_cfg', [dict(cosine=True, tau_min=0.01), dict(cosine=True, tau_min=0.01, non_shared_tau=True)])
@pytest.mark.parametrize('n_voxels', [2500, 9000])
def test_bf16_cosine_layers(layer_cfg, n_voxels):
    """scaled cosine attention in the reduced-precision mode (sst_sra_attn_cos_{fwd,bwd}_bf16: normalisation and 1 / clamp(tau)
    inside the bf16 kernels): the stack runs in bf16 (not the fp32 fallback), agrees with the fp32 cosine stack at bf16
    resolution - outputs, input gradient, every parameter gradient incl. tau - and the one-call layer executor reproduces the
    Python launch sequence bit for bit"""
    import math
    import sst_amd
    from sst_amd import bf16 as B
    torch.manual_seed(5)
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2,
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False,
                                      layer_cfg=layer_cfg)).to(DEV).train()
    with torch.no_grad():
        for i, blk in enumerate(net.block_list):
            for j, enc in enumerate(blk.encoder_list):
                enc.win_attn.self_attn.tau.fill_(0.3 + 0.2 * j)
    g = torch.Generator().manual_seed(3)
    side = int(math.ceil(math.sqrt(n_voxels * 2.2)))
    cells = torch.randperm(side * side, generator=g)[:n_voxels].sort()[0]
    coors = torch.stack([torch.zeros_like(cells), torch.zeros_like(cells), cells // side + 20, cells % side + 20], 1).to(DEV)
    feats0 = torch.randn(n_voxels, 128, generator=g).to(DEV)
    up = torch.randn(n_voxels, 128, generator=g).to(DEV)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True).eval()

    def step():
        net.zero_grad(set_to_none=True)
        feats = feats0.clone().requires_grad_(True)
        out = net(layer(feats, coors, 1))[0]['voxel_feats']
        (out * up[:out.size(0)]).sum().backward()
        return out.detach().clone(), feats.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    ref = step()                       # fp32 storage (exact-split products), cosine inside the fp32 kernels
    net.set_precision('bf16')
    calls = []
    orig = B.run_encoder_stack
    B.run_encoder_stack = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        low = step()
        B._LAYER_EXEC = 0
        seq = step()
    finally:
        B.run_encoder_stack = orig
        B._LAYER_EXEC = 1
        net.set_precision('f32x6')
    assert len(calls) == 2, 'the cosine stack did not run in the reduced-precision mode'
    assert torch.equal(low[0], seq[0]) and torch.equal(low[1], seq[1])
    for n in low[2]:
        assert torch.equal(low[2][n], seq[2][n]), n
    d = (low[0] - ref[0]).abs()
    assert float(d.max()) < 8e-2 and float(d.mean()) < 8e-3, (float(d.max()), float(d.mean()))
    sc = max(1.0, float(ref[1].abs().max()))
    e = (low[1] - ref[1]).abs()
    assert float(e.max()) < 8e-2 * sc and float(e.mean()) < 8e-3 * sc
    assert low[2].keys() == ref[2].keys() and any(n.endswith('tau') for n in low[2])
    for n in low[2]:
        s_ = max(1.0, float(ref[2][n].abs().max()))
        assert float((low[2][n] - ref[2][n]).abs().max()) < 6e-2 * s_, n
