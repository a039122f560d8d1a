"""GPU: the row-wise / dense pieces of an encoder layer against a plain PyTorch fp32 reference of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True)
def _fp32_matrix_pipe():
    """this file pins the kernels of the fp32 matrix pipe (csrc/dense_f32.hip, csrc/wgrad.hip: the opt-out mode since round 5,
    when the exact split 'f32x6' became the process default); tests that name a mode set it themselves"""
    from sst_amd import dense as D
    D.set_matmul_mode('f32')
    yield
    D.set_matmul_mode(D.DEFAULT_MATMUL_MODE)


@pytest.mark.parametrize('m,c', [(1, 128), (77, 128), (5000, 192), (90107, 128), (1000, 64), (300, 512), (18443, 133), (50000, 148), (7, 1),
                                 (333, 16), (1000, 511)])
@pytest.mark.parametrize('with_res', [True, False])
def test_add_layer_norm_matches_torch(m, c, with_res):
    from sst_amd.dense import add_layer_norm
    g = torch.Generator().manual_seed(m + c)
    x = (torch.randn(m, c, generator=g) * 2 + 0.5).to(DEV)
    r = torch.randn(m, c, generator=g).to(DEV) if with_res else None
    norm = torch.nn.LayerNorm(c).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(c, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(c, generator=g) * 0.1)
    gout = torch.randn(m, c, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ra = r.clone().requires_grad_(True) if with_res else None
    rb = r.clone().requires_grad_(True) if with_res else None
    y = add_layer_norm(xa, ra, norm)
    (y * gout).sum().backward()
    gw, gb = norm.weight.grad.clone(), norm.bias.grad.clone()
    norm.zero_grad()
    y_ref = F.layer_norm(xb + rb if with_res else xb, (c,), norm.weight, norm.bias, norm.eps)
    (y_ref * gout).sum().backward()
    assert torch.allclose(y, y_ref, atol=2e-5, rtol=1e-5)
    assert torch.allclose(xa.grad, xb.grad, atol=2e-4, rtol=1e-4)
    if with_res:
        assert torch.allclose(ra.grad, rb.grad, atol=2e-4, rtol=1e-4)
    scale = max(1.0, float(norm.weight.grad.abs().max()))
    assert torch.allclose(gw, norm.weight.grad, atol=2e-4 * scale, rtol=1e-3)
    assert torch.allclose(gb, norm.bias.grad, atol=2e-4 * scale, rtol=1e-3)


@pytest.mark.parametrize('m,c', [(18443, 128), (5000, 133), (1000, 16), (300, 512), (9, 3)])
@pytest.mark.parametrize('act', ['gelu', 'relu'])
def test_layer_norm_with_folded_activation_matches_torch(m, c, act):
    """act(LayerNorm(x)) in one pass (Linear -> LN -> GELU of FSD's SIR layers, voxel_encoder.py:628-650): output and every
    gradient against torch's composition in float64; then an MLP stage of build_mlp (sst_ops.py:334-361) end to end."""
    from sst_amd.dense import add_layer_norm
    g = torch.Generator().manual_seed(m + c)
    x = (torch.randn(m, c, generator=g) * 2 + 0.5).to(DEV)
    norm = torch.nn.LayerNorm(c).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(c, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(c, generator=g) * 0.1)
    mod = torch.nn.GELU() if act == 'gelu' else torch.nn.ReLU()
    gout = torch.randn(m, c, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.double().clone().requires_grad_(True)
    y = add_layer_norm(xa, None, norm, act=mod)
    (y * gout).sum().backward()
    gw, gb = norm.weight.grad.clone(), norm.bias.grad.clone()
    pre = F.layer_norm(xb, (c,), norm.weight.double(), norm.bias.double(), norm.eps)
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    pre = F.layer_norm(xb, (c,), wd, bd, norm.eps)
    y_ref = F.gelu(pre) if act == 'gelu' else F.relu(pre)
    (y_ref * gout.double()).sum().backward()
    assert float((y.double() - y_ref).abs().max()) < 2e-5
    safe = (pre.detach().abs() > 1e-4) if act == 'relu' else torch.ones_like(pre, dtype=torch.bool)
    assert float(((xa.grad.double() - xb.grad).abs() * safe.any(1, keepdim=True)).max()) < 5e-4 * max(1.0, float(xb.grad.abs().max()))
    flip = float((gout.double().abs() * (~safe)).sum(0).max()) * 6.0
    assert float((gw.double() - wd.grad).abs().max()) < 2e-4 * max(1.0, float(wd.grad.abs().max())) + flip
    assert float((gb.double() - bd.grad).abs().max()) < 2e-4 * max(1.0, float(bd.grad.abs().max())) + flip


def test_mlp_stage_runs_the_fused_norm_and_keeps_the_reference_layout():
    from sst_amd.sst_ops import build_mlp
    torch.manual_seed(3)
    mlp = build_mlp(3, [16, 32, 133], dict(type='LN', eps=1e-3), act='gelu').to(DEV)
    assert [k for k in mlp.state_dict()][:3] == ['0.0.weight', '0.1.weight', '0.1.bias']
    x = torch.randn(4000, 3, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = mlp(xa)
    y.square().sum().backward()
    ga = [p.grad.clone() for p in mlp.parameters()]
    mlp.zero_grad()
    t = xb
    for stage in mlp:      # the composition the reference runs: Linear, LayerNorm, GELU one after the other
        t = F.gelu(F.layer_norm(F.linear(t, stage[0].weight), stage[1].normalized_shape, stage[1].weight, stage[1].bias, stage[1].eps))
    t.square().sum().backward()
    assert float((y - t).abs().max()) < 1e-4
    assert float((xa.grad - xb.grad).abs().max()) < 1e-3 * max(1.0, float(xb.grad.abs().max()))
    for a, p in zip(ga, mlp.parameters()):
        assert float((a - p.grad).abs().max()) < 1e-3 * max(1.0, float(p.grad.abs().max()))


@pytest.mark.parametrize('m,cin,cout,bias', [(90107, 128, 384, True), (4097, 256, 128, True), (116000, 9, 64, False),
                                             (100, 128, 128, True), (50000, 128, 256, True), (90107, 128, 128, False),
                                             (33333, 96, 160, True), (116000, 128, 128, False), (5000, 64, 32, True),
                                             # ragged channel counts of the SIR / VFE layers (masked columns in the wgrad kernel)
                                             (30000, 84, 128, True), (30000, 133, 128, False), (30000, 3, 16, True),
                                             (30001, 32, 84, True), (8000, 1, 1, True), (30000, 261, 130, True)])
def test_tall_linear_matches_torch(m, cin, cout, bias):
    from sst_amd.dense import tall_linear
    g = torch.Generator().manual_seed(m)
    x = torch.randn(m, cin, generator=g).to(DEV)
    w = (torch.randn(cout, cin, generator=g) * 0.1).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV) if bias else None
    gout = (torch.randn(m, cout, generator=g) * 0.1).to(DEV)
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ba = b.clone().requires_grad_(True) if bias else None
    xb, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bb = b.clone().requires_grad_(True) if bias else None
    y = tall_linear(xa, wa, ba)
    (y * gout).sum().backward()
    y_ref = F.linear(xb.double(), wb.double(), bb.double() if bias else None)
    (y_ref * gout.double()).sum().backward()
    assert torch.allclose(y.double(), y_ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(xa.grad.double(), xb.grad.double(), atol=1e-4, rtol=1e-4)
    gs = float(wb.grad.abs().max())
    assert float((wa.grad.double() - wb.grad.double()).abs().max()) < 2e-4 * max(1.0, gs)
    if bias:
        assert float((ba.grad.double() - bb.grad.double()).abs().max()) < 2e-4 * max(1.0, float(bb.grad.abs().max()))


@pytest.mark.parametrize('m,out,inn', [(90107, 256, 128), (90107, 128, 256), (90107, 384, 128), (4096, 128, 64),
                                       (5001, 256, 256), (40000, 512, 64), (9000, 1024, 256), (70001, 128, 128)])
def test_weight_bias_grad_tiled_shapes(m, out, inn):
    """csrc/wgrad.hip tiled mode: one workgroup per 128 x 64 tile of dW and K slice, any tile count up to 64,
    ragged K chunks (m not a multiple of the slice length)."""
    from sst_amd.dense import weight_bias_grad
    g = torch.Generator().manual_seed(m + out)
    dy = (torch.randn(m, out, generator=g) * 0.1).to(DEV)
    x = torch.randn(m, inn, generator=g).to(DEV)
    dw, db = weight_bias_grad(dy, x, True)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    assert float((dw.double() - ref_w).abs().max()) < 2e-4 * max(1.0, float(ref_w.abs().max()))
    assert float((db.double() - ref_b).abs().max()) < 2e-4 * max(1.0, float(ref_b.abs().max()))
    # views with a row stride (column slices of a packed buffer), no bias
    big_dy = (torch.randn(m, out + 64, generator=g) * 0.1).to(DEV)
    big_x = torch.randn(m, inn + 32, generator=g).to(DEV)
    dw2, db2 = weight_bias_grad(big_dy[:, 64:], big_x[:, :inn], False)
    ref2 = big_dy[:, 64:].double().t() @ big_x[:, :inn].double()
    assert db2 is None
    assert float((dw2.double() - ref2).abs().max()) < 2e-4 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize('m,c', [(1, 4), (1000, 128), (90107, 384), (33333, 256), (5000, 1024)])
def test_colsum(m, c):
    from sst_amd.dense import colsum
    g = torch.Generator().manual_seed(c)
    x = torch.randn(m, c, generator=g).to(DEV)
    ref = x.double().sum(0)
    out = colsum(x)
    assert float((out.double() - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))
    xs = torch.randn(m, 2 * c, generator=g).to(DEV)[:, c:]   # row-strided view
    assert float((colsum(xs).double() - xs.double().sum(0)).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('m', [1, 31, 32, 33, 1000, 8192, 90107, 262144])
@pytest.mark.parametrize('k,trans,acc', [(128, False, False), (256, False, False), (128, True, True), (256, True, True),
                                         (256, True, False)])
def test_tall_gemm_matches_float64(m, k, trans, acc):
    """csrc/tall_gemm.hip (hand-pipelined: asm-issued loads with explicit waits) against a float64 product.
    Repeated launches: a register touched while its load is still in flight would show up as run-to-run noise."""
    from sst_amd.dense import tall_gemm
    dev = torch.device('cuda:0')
    torch.manual_seed(m + k)
    x = torch.randn(m, k, device=dev)
    w = torch.randn(k, 128, device=dev) if trans else torch.randn(128, k, device=dev)
    b = torch.randn(128, device=dev)
    y0 = torch.randn(m, 128, device=dev)
    wt = w if trans else w.t()
    ref = x.double() @ wt.double() + b.double() + (y0.double() if acc else 0)
    first = None
    for rep in range(4):
        out = y0.clone() if acc else torch.full((m, 128), float('nan'), device=dev)
        got = tall_gemm(x, w, b, trans_w=trans, out=out, accumulate=acc)
        assert got is out
        assert (out.double() - ref).abs().max().item() < 1e-3
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(first, out)  # deterministic, bit for bit


@pytest.mark.parametrize('m', [1, 33, 5000, 90107])
def test_gelu_fused_gemms_match_float64(m):
    """linear1 + bias + GELU in one pass and (dy w2) * gelu'(pre) in one pass (csrc/tall_gemm.hip epilogues) against
    the float64 composition of torch's erf GELU (sst_basic_block_v2.py:116)."""
    from sst_amd.dense import dgrad_gelu, linear_gelu
    dev = torch.device('cuda:0')
    torch.manual_seed(m)
    x = torch.randn(m, 128, device=dev)
    w1 = torch.randn(256, 128, device=dev) * 0.1
    b1 = torch.randn(256, device=dev)
    w2 = torch.randn(128, 256, device=dev) * 0.1
    dy = torch.randn(m, 128, device=dev)
    for rep in range(3):
        pre, h = linear_gelu(x, w1, b1)
        pre_ref = x.double() @ w1.double().t() + b1.double()
        assert (pre.double() - pre_ref).abs().max().item() < 1e-4
        assert (h.double() - torch.nn.functional.gelu(pre_ref)).abs().max().item() < 1e-4
        dpre = dgrad_gelu(dy, w2, pre)
        p = pre.double().requires_grad_(True)
        (torch.nn.functional.gelu(p) * (dy.double() @ w2.double())).sum().backward()
        assert (dpre.double() - p.grad).abs().max().item() < 1e-4
    assert linear_gelu(torch.randn(8, 64, device=dev), torch.randn(256, 64, device=dev), b1) is None


def test_tall_gemm_strided_operands_and_unsupported_shapes():
    from sst_amd.dense import tall_gemm
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    big = torch.randn(5000, 384, device=dev)
    w_in = torch.randn(384, 128, device=dev)
    x = big[:, 128:256]                      # row stride 384
    y = tall_gemm(x, w_in[256:], None)       # row slice of a packed in-projection weight
    assert (y.double() - x.double() @ w_in[256:].double().t()).abs().max().item() < 1e-3
    assert tall_gemm(torch.randn(100, 64, device=dev), torch.randn(128, 64, device=dev)) is None   # K = 64
    assert tall_gemm(torch.randn(100, 128, device=dev), torch.randn(256, 128, device=dev)) is None  # N = 256


@pytest.mark.parametrize('m', [1, 77, 5000, 90107])
@pytest.mark.parametrize('k,n', [(128, 128), (128, 256), (256, 128)])
@pytest.mark.parametrize('trans_w', [False, True])
def test_lds_linear_f32(m, k, n, trans_w):
    """csrc/dense_f32.hip against float64 on the same operands: exact-fp32 products, every epilogue, both weight layouts"""
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(m + k + 3 * n + int(trans_w))
    x = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
    wa = w.t().contiguous() if trans_w else w
    b = torch.randn(n, generator=g).to(DEV)
    aux = torch.randn(m, n, generator=g).to(DEV)
    ref = (x.double() @ w.double().t())
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert D.lds_linear_ok(x, wa, trans_w)
    y = D.lds_linear(x, wa, b, trans_w=trans_w)
    assert float((y.double() - (ref + b.double())).abs().max()) < tol
    y = D.lds_linear(x, wa, None, trans_w=trans_w)
    assert float((y.double() - ref).abs().max()) < tol
    for epi, fn in ((D.EPI_GELU, torch.nn.functional.gelu), (D.EPI_RELU, torch.relu)):
        y, pre = D.lds_linear(x, wa, b, epi, trans_w=trans_w, want_pre=True)
        assert float((pre.double() - (ref + b.double())).abs().max()) < tol
        assert float((y.double() - fn(ref + b.double())).abs().max()) < tol
    xg = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(xg).sum().backward()
    y = D.lds_linear(x, wa, None, D.EPI_MUL_GELU_GRAD, trans_w=trans_w, aux_in=aux)
    assert float((y.double() - ref * xg.grad).abs().max()) < tol
    y = D.lds_linear(x, wa, None, D.EPI_MUL_RELU_GRAD, trans_w=trans_w, aux_in=aux)
    assert float((y.double() - ref * (aux.double() > 0)).abs().max()) < tol
    y = D.lds_linear(x, wa, None, D.EPI_ADD, trans_w=trans_w, aux_in=aux)
    assert float((y.double() - (ref + aux.double())).abs().max()) < tol
    acc = aux.clone()
    D.lds_linear(x, wa, None, D.EPI_ADD, trans_w=trans_w, aux_in=acc, out=acc)      # in place: acc += x w^T
    assert float((acc.double() - (ref + aux.double())).abs().max()) < tol
    # column slices of a wider tensor as operands (dq | dk of the [M, 3C] gradient buffer, rows of in_proj_weight)
    wide = torch.randn(m, k + 64, generator=g).to(DEV)
    y = D.lds_linear(wide[:, 64:], wa, None, trans_w=trans_w)
    assert float((y.double() - wide[:, 64:].double() @ w.double().t()).abs().max()) < tol


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'f32x6'])
@pytest.mark.parametrize('m', [2500, 20000])
def test_weight_and_bias_gradients_bit_reproducible(mode, m):
    """VERDICT round 3 (cross-workgroup hand-overs): 200 launches of the grouped weight / bias gradients on the same operands
    give the same bits - below 4 096 rows the per-problem path (split-K + column sums: csrc/wgrad.hip, colsum_k), above it the
    grouped kernels (wgrad_wide_k + wgrad_reduce_group_k, or wgrad_x6_k + wgrad_x6_reduce_k).  The column-sum / LayerNorm
    backward kernels used float atomics into LDS and the output until round 4: last-bit differences from launch to launch."""
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(m)
    dqkv, ds1 = torch.randn(m, 384, generator=g).to(DEV), torch.randn(m, 128, generator=g).to(DEV)
    o, xp, x = (torch.randn(m, 128, generator=g).to(DEV) for _ in range(3))
    D.set_matmul_mode(mode)
    try:
        ref = None
        for _ in range(200):
            probs = [(ds1, o, torch.empty(128, 128, device=DEV), torch.empty(128, device=DEV)),
                     (dqkv[:, :256], xp, torch.empty(256, 128, device=DEV), torch.empty(256, device=DEV)),
                     (dqkv[:, 256:], x, torch.empty(128, 128, device=DEV), torch.empty(128, device=DEV))]
            D.weight_bias_grad_group(probs)
            cur = [t.clone() for p in probs for t in p[2:]]
            if ref is None:
                ref = cur
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))
    finally:
        D.set_matmul_mode(D.DEFAULT_MATMUL_MODE)


@pytest.mark.gpu
@pytest.mark.parametrize('c', [4, 64, 128, 133, 256])
def test_layernorm_backward_bit_reproducible(c):
    from sst_amd import dense as D
    g = torch.Generator().manual_seed(c)
    m = 30000
    s, dy = torch.randn(m, c, generator=g).to(DEV), torch.randn(m, c, generator=g).to(DEV)
    w, b = torch.randn(c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
    _, _, stats = D.add_ln_fwd(s, None, w, b, 1e-5)
    ref = None
    for _ in range(100):
        cur = [t.clone() for t in D.add_ln_act_bwd(dy, s, stats, w, b, 'gelu')]
        cur += [t.clone() for t in D.add_ln_bwd(dy, s, stats, w)]
        if ref is None:
            ref = cur
        assert all(torch.equal(a, b_) for a, b_ in zip(ref, cur))
    if c % 4 == 0:
        r0 = D.colsum(dy)
        for _ in range(100):
            assert torch.equal(r0, D.colsum(dy))
