"""GPU: the exact-split linears (csrc/dense_f32x6.hip: x = x0 + x1 + x2 and w = w0 + w1 + w2 in bf16, the six products x_i w_j
with i + j <= 2 on the bf16 matrix pipe, fp32 accumulation) against float64, BESIDE the native fp32-MFMA kernels of
csrc/dense_f32.hip on the same inputs.

The admissibility bar: the mode may carry the headline only if its largest error against the float64
oracle is <= 2 x that of the native fp32 kernel on the same inputs, for every shape / orientation / epilogue of the linears and
through the 12-layer encoder stack.  That is what these tests assert (plus an absolute bar of fp32-rounding class, 2e-6 of the
output scale), so that a change that breaks the property breaks the suite, not just a number in a report."""
import numpy as np
import pytest
import torch

from conftest import DROP_TEST

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SHAPES = [(128, 128), (128, 256), (256, 128)]


def _both(D, fn):
    """fn() under the exact-fp32 kernels and under the exact split"""
    out = {}
    try:
        for mode in ('f32', 'f32x6'):
            D.set_matmul_mode(mode)
            out[mode] = fn()
    finally:
        D.set_matmul_mode(D.DEFAULT_MATMUL_MODE)
    return out['f32'], out['f32x6']


def _admissible(native, split, want, what, scale=None, floor=2e-7):
    """max |split - float64| <= 2 x max |native - float64| (with a floor of a fraction of an fp32 rounding step of the output
    scale: at M = 1 either error can be 0)"""
    scale = max(1.0, float(want.abs().max())) if scale is None else scale
    e_n = float((native.double() - want).abs().max())
    e_s = float((split.double() - want).abs().max())
    assert e_s <= max(2.0 * e_n, floor * scale), (what, e_s, e_n, scale)
    assert e_s <= 2e-6 * scale, (what, e_s, scale)
    return e_s, e_n


@pytest.mark.parametrize('m', [1, 77, 5000, 90107])
@pytest.mark.parametrize('k,n', SHAPES)
@pytest.mark.parametrize('trans_w', [False, True])
def test_linear_all_epilogues_admissible(m, k, n, trans_w):
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(m + k + 2 * n + int(trans_w))
    x = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn((k, n) if trans_w else (n, k), generator=g) * 0.2).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    aux = torch.randn(m, n, generator=g).to(DEV)
    ref = x.double() @ (w.double() if trans_w else w.double().t()) + b.double()
    ad = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(ad).sum().backward()
    cases = [('bias', D.EPI_BIAS, None, ref), ('mul_relu_grad', D.EPI_MUL_RELU_GRAD, aux, ref * (aux.double() > 0)),
             ('add', D.EPI_ADD, aux, ref + aux.double()), ('mul_gelu_grad', D.EPI_MUL_GELU_GRAD, aux, ref * ad.grad)]
    for name, epi, a, want in cases:
        nat, spl = _both(D, lambda: D.lds_linear(x, w, b, epi, trans_w, aux_in=a))
        # the erf approximation of the epilogue (1.5e-7 absolute) sits on top of both kernels alike
        _admissible(nat, spl, want, name, floor=1e-6 if 'gelu' in name else 2e-7)
        if name == 'bias' and m >= 77:
            assert not torch.equal(nat, spl)            # the split kernel really ran
    for name, epi, act in (('gelu', D.EPI_GELU, torch.nn.functional.gelu), ('relu', D.EPI_RELU, torch.relu)):
        (yn, pn), (ys, ps) = _both(D, lambda: D.lds_linear(x, w, b, epi, trans_w, want_pre=True))
        _admissible(pn, ps, ref, name + ' pre-activation')
        _admissible(yn, ys, act(ref), name, floor=1e-6)


@pytest.mark.parametrize('m', [1, 77, 5000, 90107])
def test_linear_add_layernorm_admissible(m):
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(m)
    x = torch.randn(m, 128, generator=g).to(DEV)
    w = (torch.randn(128, 128, generator=g) * 0.2).to(DEV)
    b, lw, lb = (torch.randn(128, generator=g).to(DEV) for _ in range(3))
    res = torch.randn(m, 128, generator=g).to(DEV)
    table = torch.randn(144, 128, generator=g).to(DEV)
    idx = torch.randint(0, 144, (m,), generator=g, dtype=torch.int32).to(DEV)
    nat, spl = _both(D, lambda: D.lds_linear_add_ln(x, w, b, res, lw, lb, 1e-5, pos=(table, idx)))
    ssum = x.double() @ w.double().t() + b.double() + res.double()
    ref = torch.nn.functional.layer_norm(ssum, (128,), lw.double(), lb.double(), 1e-5)
    _admissible(nat[1], spl[1], ssum, 'sum')
    _admissible(nat[0], spl[0], ref, 'layer norm', floor=1e-6)
    _admissible(nat[3], spl[3], ref + table.double()[idx.long()], 'layer norm + pos', floor=1e-6)
    # K = 256 has no split kernel with a fused LayerNorm (three weight images of a 128-column group exceed the LDS): the C entry
    # refuses it, the Python side routes that one product to the fp32-pipe kernel
    from sst_amd import _lib
    rc = _lib.load().sst_tall_linear_ln_f32x6(_lib.ptr(x), 256, _lib.ptr(w), 256, _lib.ptr(b), m, 256, _lib.ptr(res), 128, _lib.ptr(lw),
                                              _lib.ptr(lb), 1e-5, _lib.ptr(nat[0]), None, _lib.ptr(nat[2]), None, None, None, None)
    assert rc != 0


@pytest.mark.parametrize('m', [1, 300, 20000])
def test_qkv_one_launch_equals_the_two_projections(m):
    """q | k = (x + pos) W_qk, v = x W_v as ONE launch over N = 384 with two input matrices: bit-identical to the separate
    launches of the same kernel (same products in the same order per output element)"""
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(5 * m)
    x = torch.randn(m, 128, generator=g).to(DEV)
    xp = x + torch.randn(m, 128, generator=g).to(DEV)
    w = (torch.randn(384, 128, generator=g) * 0.2).to(DEV)
    b = torch.randn(384, generator=g).to(DEV)
    D.set_matmul_mode('f32x6')
    try:
        assert D.lds_linear_qkv_ok(xp, x, w)
        one = D.lds_linear_qkv(xp, x, w, b)
        qk = D.lds_linear(xp, w[:256], b[:256])
        v = D.lds_linear(x, w[256:], b[256:])
    finally:
        D.set_matmul_mode(D.DEFAULT_MATMUL_MODE)
    assert torch.equal(one[:, :256], qk) and torch.equal(one[:, 256:], v)
    want = torch.cat([xp.double() @ w[:256].double().t(), x.double() @ w[256:].double().t()], 1) + b.double()
    assert float((one.double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    with D.matmul_mode_scope('f32'):
        assert not D.lds_linear_qkv_ok(xp, x, w)       # outside the mode: the caller keeps its two launches


def _stack_case(n_voxels, seed):
    """a 12-layer SSTv2 stack (6 shift blocks, d 128, 8 heads, FFN 256) with xavier weights and non-trivial norms / biases on a
    random voxel set; -> (net, voxel_info, float64 oracle output)"""
    import sst_amd
    from oracle import sst_oracle
    torch.manual_seed(seed)
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128] * 6, nhead=[8] * 6, num_blocks=6, dim_feedforward=[256] * 6,
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    net = net.to(DEV).eval()
    side = 96
    cells = torch.randperm(side * side, generator=g)[:n_voxels].sort()[0]
    coors = torch.stack([torch.zeros_like(cells), torch.zeros_like(cells), cells // side + 100, cells % side + 100], 1).to(DEV)
    feats = torch.randn(n_voxels, 128, generator=g).to(DEV)
    layer = sst_amd.SSTInputLayerV2((DROP_TEST, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True, reference_outputs=False).eval()
    info = layer(feats, coors, 1)
    x = info['voxel_feats'].double().cpu().numpy()
    layers = [enc for block in net.block_list for enc in block.encoder_list]
    for li, enc in enumerate(layers):
        plan = info[f'sra_plan_shift{li % 2}']
        pos = info[f'pos_embed_shift{li % 2}'].double().cpu().numpy()
        params = {k: v.detach().double().cpu().numpy() for k, v in enc.state_dict().items()}
        x = sst_oracle.encoder_layer(x, pos, plan.tok.cpu().numpy(), plan.winoff[:plan.n_windows + 1].cpu().numpy(), params, 8)
    return net, info, torch.from_numpy(x)


@pytest.mark.parametrize('n_voxels,seed', [(3000, 0), (6000, 3)])
def test_twelve_layer_stack_admissible(n_voxels, seed):
    """the whole encoder stack, forward: the exact split against float64 (oracle/sst_oracle.encoder_layer x 12, numpy float64)
    beside the native fp32 mode on the same weights, plans and features"""
    net, info, want = _stack_case(n_voxels, seed)
    outs = {}
    try:
        for mode in ('fp32', 'f32x6'):
            net.set_precision(mode)
            with torch.no_grad():
                outs[mode] = net(dict(info))[0]['voxel_feats'].cpu()
    finally:
        net.set_precision('fp32')
    scale = max(1.0, float(want.abs().max()))
    e_n = float((outs['fp32'].double() - want).abs().max())
    e_s = float((outs['f32x6'].double() - want).abs().max())
    # after 12 layers of LayerNorm-ed features both sit at a few 1e-6 of float64; the bar: not worse than twice the native kernels
    assert e_s <= max(2.0 * e_n, 1e-6 * scale), (e_s, e_n)
    assert e_n <= 5e-5 * scale and e_s <= 5e-5 * scale, (e_s, e_n)
    assert not torch.equal(outs['fp32'], outs['f32x6'])


def test_sst_block_f32x6_training_step_vs_exact():
    """two encoder layers forward + backward through FusedEncoderLayerFn in the mode (q | k | v as one launch): outputs, input gradient and every parameter gradient against the exact-fp32 mode"""
    import sst_amd
    from conftest import DROP_TRAIN, load_golden
    g = load_golden('sst_block_std.npz')
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=True))
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}, strict=True)
    net = net.to(DEV).train()
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False,
                                    debug=True, mute=True, reference_outputs=False)
    layer.eval()
    coors = torch.from_numpy(g['in::voxel_coors']).to(DEV)
    up = torch.from_numpy(g['in::grad_out']).to(DEV)
    outs = {}
    try:
        for mode in ('fp32', 'f32x6'):
            net.set_precision(mode)
            net.zero_grad(set_to_none=True)
            feats = torch.from_numpy(g['in::voxel_feats']).to(DEV).requires_grad_(True)
            out = net(layer(feats, coors, 2))[0]['voxel_feats']
            (out * up).sum().backward()
            outs[mode] = (out.detach().clone(), feats.grad.clone(),
                          {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    finally:
        net.set_precision('fp32')
    (o32, g32, p32), (o6, g6, p6) = outs['fp32'], outs['f32x6']
    assert float((o32 - o6).abs().max()) <= 5e-6 and not torch.equal(o32, o6)
    assert np.abs(o6.cpu().numpy() - g['out::voxel_feats']).max() <= 2e-5          # the reference's own golden
    sc = max(1.0, float(g32.abs().max()))
    assert float((g32 - g6).abs().max()) <= 5e-6 * sc
    for name, grad in p6.items():
        s_ = max(1.0, float(p32[name].abs().max()))
        assert float((grad - p32[name]).abs().max()) <= 1e-5 * s_, name


@pytest.mark.parametrize('m', [4096, 5001, 90107])
def test_weight_gradients_admissible(m):
    """the five parameter gradients of an encoder layer as the backward pass groups them (csrc/wgrad_x6.hip): against float64,
    beside the fp32-pipe kernel (csrc/wgrad.hip) on the same operands - strided views of a [M, 384] buffer included"""
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(m)
    dev = DEV
    dqkv = torch.randn(m, 384, generator=g).to(dev)
    ds2, ds1 = torch.randn(m, 128, generator=g).to(dev), torch.randn(m, 128, generator=g).to(dev)
    dpre = torch.randn(m, 256, generator=g).to(dev)
    h = torch.randn(m, 256, generator=g).to(dev)
    y1, o, xp, x = (torch.randn(m, 128, generator=g).to(dev) for _ in range(4))
    groups = [[(ds2, h), (dpre, y1)], [(ds1, o), (dqkv[:, :256], xp), (dqkv[:, 256:], x)]]

    def run():
        outs = []
        for grp in groups:
            probs = [(dy, xx, torch.empty(dy.size(1), xx.size(1), device=dev), torch.empty(dy.size(1), device=dev)) for dy, xx in grp]
            D.weight_bias_grad_group(probs)
            outs += [(p[2], p[3]) for p in probs]
        return outs

    nat, spl = _both(D, run)
    flat = [p for grp in groups for p in grp]
    for (dy, xx), (wn, bn), (ws, bs) in zip(flat, nat, spl):
        want_w = dy.double().t() @ xx.double()
        want_b = dy.double().sum(0)
        # the reference scale of a contraction over m tokens: sum |dy| |x| ~ m
        _admissible(wn, ws, want_w, 'dW %s' % (tuple(want_w.shape),), floor=2e-7)
        _admissible(bn, bs, want_b, 'db', floor=1e-6)
        assert not torch.equal(wn, ws)


def test_weight_gradients_small_and_unsupported_shapes_fall_back():
    """fewer than 4096 rows or widths that are not multiples of 128: the exact-fp32 kernels (same results as outside the mode)"""
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(3)
    dy, x = torch.randn(5000, 64, generator=g).to(DEV), torch.randn(5000, 128, generator=g).to(DEV)

    def run():
        w, b = torch.empty(64, 128, device=DEV), torch.empty(64, device=DEV)
        D.weight_bias_grad_group([(dy, x, w, b)])
        return w, b

    (wn, bn), (ws, bs) = _both(D, run)
    assert torch.equal(wn, ws) and torch.equal(bn, bs)
    assert float((wn.double() - dy.double().t() @ x.double()).abs().max()) <= 1e-3


@pytest.mark.parametrize('m', [77, 20000, 90107])
def test_in_projection_data_gradient_as_one_product(m):
    """d(x) = ds1 + [dq | dk | dv] in_proj_weight over K = 384 in one launch (residual in the epilogue): against float64 and
    beside the two fp32-pipe products it replaces"""
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(m)
    dqkv = torch.randn(m, 384, generator=g).to(DEV)
    w = (torch.randn(384, 128, generator=g) * 0.2).to(DEV)
    ds1 = torch.randn(m, 128, generator=g).to(DEV)
    want = ds1.double() + dqkv.double() @ w.double()
    D.set_matmul_mode('f32x6')
    try:
        assert D.lds_linear_dqkv_ok(dqkv, w)
        out = ds1.clone()
        got = D.lds_linear(dqkv, w, None, D.EPI_ADD, trans_w=True, aux_in=out, out=out)
    finally:
        D.set_matmul_mode(D.DEFAULT_MATMUL_MODE)
    with D.matmul_mode_scope('f32'):
        assert not D.lds_linear_dqkv_ok(dqkv, w)
        nat = ds1.clone()
        D.lds_linear(dqkv[:, :256].contiguous(), w[:256], None, D.EPI_ADD, trans_w=True, aux_in=nat, out=nat)
        D.lds_linear(dqkv[:, 256:].contiguous(), w[256:], None, D.EPI_ADD, trans_w=True, aux_in=nat, out=nat)
    _admissible(nat, got, want, 'K = 384 data gradient')
