"""GPU: sparse 3-D convolution (csrc/spconv.hip, sst_amd/spconv.py).  Rulebooks: bit-exact against the oracle
(oracle/spconv_oracle.py) and, through coordinates, against the output of the reference's own CPU templates
(tests/golden/spconv.npz).  Convolution forward / data gradient / weight gradient: against the float64 restatement
of indiceConv / indiceConvBackward, relative 1e-4 (fp32 MFMA accumulation)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle import _rulebook_equal_by_coordinates, spconv_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TAGS = ['subm3', 'down3s2', 'down_k313', 'down2s2', 'transposed', 'subm_dil2']


def _cloud(rng, n, batch, shape):
    vol = int(np.prod(shape))
    lin = rng.choice(batch * vol, n, replace=False)
    b, r = lin // vol, lin % vol
    return np.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]], 1).astype(np.int32)


def _rulebook(ind, batch, shape, ks, st, pd, dl, subm, tr):
    from sst_amd import spconv
    outids, pairs, num = spconv.get_indice_pairs(torch.from_numpy(ind).to(DEV), batch, shape, ks, st, pd, dl, 0, subm, tr)
    return outids, pairs, num, pairs._sst_rulebook


@pytest.mark.parametrize('builder', ['grid', 'sort'])
@pytest.mark.parametrize('tag', TAGS)
def test_rulebook_matches_oracle_and_reference_golden(tag, builder, monkeypatch):
    """both rulebook builders (dense cell grid; sort + binary search) against the oracle and the reference's own output"""
    from oracle import spconv_oracle as O
    monkeypatch.setenv('SST_SPCONV_RULEBOOK', builder)
    g = load_golden('spconv.npz')
    ind, batch, shape, ks, st, pd, dl, subm, tr = spconv_case(g, tag)
    outids, pairs, num, rb = _rulebook(ind, batch, shape, ks, st, pd, dl, subm, tr)
    w_out, w_pairs, w_num, _ = O.indice_pairs(ind, batch, shape, ks, st, pd, dl, (0, 0, 0), subm, tr)
    assert outids.dtype == torch.int32 and pairs.dtype == torch.int32 and num.dtype == torch.int32
    np.testing.assert_array_equal(outids.cpu().numpy(), w_out)       # sorted (b, z, y, x) / the inputs for subm
    np.testing.assert_array_equal(num.cpu().numpy(), w_num)
    np.testing.assert_array_equal(pairs.cpu().numpy(), w_pairs)      # pairs by ascending input row, -1 behind
    in2out, out2in = O.maps_from_pairs(w_pairs, w_num, len(ind), len(w_out))
    np.testing.assert_array_equal(rb.in2out.cpu().numpy(), in2out)
    np.testing.assert_array_equal(rb.out2in.cpu().numpy(), out2in)
    _rulebook_equal_by_coordinates(g[f'out::{tag}::outids'], g[f'out::{tag}::pairs'], g[f'out::{tag}::num'],
                                   outids.cpu().numpy(), pairs.cpu().numpy(), num.cpu().numpy())


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('cin,cout', [(16, 32), (5, 7), (64, 64), (67, 128), (128, 160)])
def test_conv_forward_backward_match_oracle(tag, cin, cout):
    from oracle import spconv_oracle as O
    from sst_amd import spconv
    g = load_golden('spconv.npz')
    ind, batch, shape, ks, st, pd, dl, subm, tr = spconv_case(g, tag)
    outids, pairs, num, rb = _rulebook(ind, batch, shape, ks, st, pd, dl, subm, tr)
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(len(ind), cin, generator=gen)
    w = torch.randn(*ks, cin, cout, generator=gen) * 0.2
    gy = torch.randn(len(outids), cout, generator=gen)
    fn = spconv.indice_subm_conv if subm else spconv.indice_conv_fn
    xa, wa = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = fn(xa, wa, pairs, num, len(outids))
    y.backward(gy.to(DEV))
    p_np, n_np = pairs.cpu().numpy(), num.cpu().numpy()
    y_ref = O.indice_conv(x.numpy(), w.numpy(), p_np, n_np, len(outids))
    dx_ref, dw_ref = O.indice_conv_backward(x.numpy(), w.numpy(), gy.numpy(), p_np, n_np)
    for got, want in ((y, y_ref), (xa.grad, dx_ref), (wa.grad, dw_ref)):
        err = np.abs(got.detach().cpu().numpy().astype(np.float64) - want).max()
        assert err < 1e-4 * max(1.0, np.abs(want).max()), err


@pytest.mark.parametrize('tile_cfg', [41, 42, 81, 82])
@pytest.mark.parametrize('cin,cout', [(64, 64), (16, 32), (128, 160), (192, 256)])
def test_output_stationary_kernel_every_tile_shape(tile_cfg, cin, cout):
    """csrc/spconv_os.hip: each of the four tile shapes (64 / 128 rows x 64 / 128 columns), forward orientation and the
    transposed-weight orientation of the data gradient, several row tiles with a ragged last one, against the float64
    restatement of indiceConv (spconv_ops.h:256-357)."""
    from oracle import spconv_oracle as O
    from sst_amd import spconv
    rng = np.random.default_rng(cin + cout + tile_cfg)
    batch, shape, n = 2, [6, 24, 26], 1500
    ind = _cloud(rng, n, batch, shape)
    for subm, st in ((True, 1), (False, 2)):
        outids, pairs, num, rb = _rulebook(ind, batch, shape, [3] * 3, [st] * 3, [1] * 3, [1] * 3, subm, False)
        m = len(outids)
        gen = torch.Generator().manual_seed(7)
        x = torch.randn(n, cin, generator=gen)
        w = torch.randn(27, cin, cout, generator=gen) * 0.2
        gy = torch.randn(m, cout, generator=gen)
        p_np, n_np = pairs.cpu().numpy(), num.cpu().numpy()
        y = spconv._gather_gemm(x.to(DEV), rb.out2in, m, w.to(DEV), False, cout, rb.density, tile_cfg)
        y_ref = O.indice_conv(x.numpy(), w.numpy().reshape(3, 3, 3, cin, cout), p_np, n_np, m)
        assert np.abs(y.cpu().numpy() - y_ref).max() < 1e-4 * max(1.0, np.abs(y_ref).max())
        dx = spconv._gather_gemm(gy.to(DEV), rb.in2out, n, w.to(DEV), True, cin, rb.density, tile_cfg)
        dx_ref, _ = O.indice_conv_backward(x.numpy(), w.numpy().reshape(3, 3, 3, cin, cout), gy.numpy(), p_np, n_np)
        assert np.abs(dx.cpu().numpy() - dx_ref).max() < 1e-4 * max(1.0, np.abs(dx_ref).max())


@pytest.mark.parametrize('tile_cfg', [41, 82])
def test_output_stationary_kernel_launch_order(tile_cfg, monkeypatch):
    """the heaviest-first launch order of the row tiles (csrc/spconv_os.hip, sp_os_tile_work_k + a sort, cached on the
    rulebook): the work figures equal a numpy count of the populated (16-row block, offset) slots, the order is a
    permutation by decreasing work, and the contraction gives bit for bit what it gives in row order."""
    from sst_amd import spconv, _lib
    rng = np.random.default_rng(tile_cfg)
    batch, shape, n, cin, cout = 2, [6, 40, 44], 5000, 32, 64 if tile_cfg == 41 else 128
    ind = _cloud(rng, n, batch, shape)
    _, _, _, rb = _rulebook(ind, batch, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, False)
    lib = _lib.load()
    rows = lib.sst_spconv_conv_os_tile_rows(n, cout, tile_cfg)
    assert rows == 64 * (tile_cfg % 10)
    order = rb.tile_order(rb.out2in, n, rows)
    n_tiles = -(-n // rows)
    live = rb.out2in.cpu().numpy() >= 0
    live = np.pad(live, ((0, 0), (0, n_tiles * rows - n))).reshape(27, n_tiles, rows // 16, 16).any(-1)
    work = live.sum((0, 2))
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(n_tiles))
    assert (np.diff(work[o]) <= 0).all() and work.max() > work.min()
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, generator=gen).to(DEV)
    w = (torch.randn(27, cin, cout, generator=gen) * 0.2).to(DEV)
    y_rows = spconv._gather_gemm(x, rb.out2in, n, w, False, cout, rb.density, tile_cfg)
    monkeypatch.setattr(spconv, '_OS_ORDER_MIN_ROWS', 0)
    y_ordered = spconv._gather_gemm(x, rb.out2in, n, w, False, cout, rb, tile_cfg)
    assert torch.equal(y_rows, y_ordered)


@pytest.mark.parametrize('cin,cout', [(64, 64), (16, 32), (128, 160), (192, 256), (64, 128)])
def test_split_precision_convolution_kernel(cin, cout):
    """csrc/spconv_os_x3.hip: the contraction as three bf16 products of split fp32 operands (fp32 accumulation), forward
    orientation and the transposed-weight orientation of the data gradient, against the float64 restatement of indiceConv
    (spconv_ops.h:256-357): ~1e-5 of the output scale (the exact-fp32 kernel: ~1e-6), far inside the 1e-3 feature tolerance;
    then a submanifold layer end to end in that mode (filter gradient exact fp32)."""
    from oracle import spconv_oracle as O
    from sst_amd import spconv
    rng = np.random.default_rng(cin + cout)
    batch, shape, n = 2, [6, 40, 44], 6000
    ind = _cloud(rng, n, batch, shape)
    try:
        for subm, st in ((True, 1), (False, 2)):
            outids, pairs, num, rb = _rulebook(ind, batch, shape, [3] * 3, [st] * 3, [1] * 3, [1] * 3, subm, False)
            m = len(outids)
            gen = torch.Generator().manual_seed(7)
            x = torch.randn(n, cin, generator=gen) * 3
            w = torch.randn(27, cin, cout, generator=gen) * 0.2
            gy = torch.randn(m, cout, generator=gen)
            p_np, n_np = pairs.cpu().numpy(), num.cpu().numpy()
            y_ref = O.indice_conv(x.numpy(), w.numpy().reshape(3, 3, 3, cin, cout), p_np, n_np, m)
            dx_ref, _ = O.indice_conv_backward(x.numpy(), w.numpy().reshape(3, 3, 3, cin, cout), gy.numpy(), p_np, n_np)
            errs = {}
            for mode in ('f32', 'f32x3'):
                spconv.set_conv_precision(mode)
                y = spconv._gather_gemm(x.to(DEV), rb.out2in, m, w.to(DEV), False, cout, rb)
                dx = spconv._gather_gemm(gy.to(DEV), rb.in2out, n, w.to(DEV), True, cin, rb)
                errs[mode] = (np.abs(y.cpu().numpy() - y_ref).max() / max(1.0, np.abs(y_ref).max()),
                              np.abs(dx.cpu().numpy() - dx_ref).max() / max(1.0, np.abs(dx_ref).max()))
            assert max(errs['f32']) < 5e-6 and max(errs['f32x3']) < 5e-5, errs
    finally:
        spconv.set_conv_precision(spconv.DEFAULT_CONV_PRECISION)


@pytest.mark.parametrize('cin,cout', [(64, 64), (16, 32), (128, 160), (192, 256), (64, 128), (256, 256)])
def test_exact_split_convolution_kernel(cin, cout):
    """csrc/spconv_os_x6.hip: the contraction from the EXACT three-way bf16 split (six products, two accumulator sets), forward
    orientation and the transposed-weight orientation of the data gradient, against the float64 restatement of indiceConv
    (spconv_ops.h:256-357) BESIDE the fp32-pipe kernel on the same operands: its error may be at most twice the fp32 kernel's
    (the admissibility bar of the dense layers, tests/test_gpu_dense_f32x6.py) - then it is the same arithmetic class and may
    carry the FSD / FSDv2 lines."""
    from oracle import spconv_oracle as O
    from sst_amd import spconv
    rng = np.random.default_rng(cin + cout)
    batch, shape, n = 2, [6, 40, 44], 6000
    ind = _cloud(rng, n, batch, shape)
    try:
        for subm, st in ((True, 1), (False, 2)):
            outids, pairs, num, rb = _rulebook(ind, batch, shape, [3] * 3, [st] * 3, [1] * 3, [1] * 3, subm, False)
            m = len(outids)
            gen = torch.Generator().manual_seed(7)
            x = torch.randn(n, cin, generator=gen) * 3
            w = torch.randn(27, cin, cout, generator=gen) * 0.2
            gy = torch.randn(m, cout, generator=gen)
            p_np, n_np = pairs.cpu().numpy(), num.cpu().numpy()
            y_ref = O.indice_conv(x.numpy(), w.numpy().reshape(3, 3, 3, cin, cout), p_np, n_np, m)
            dx_ref, dw_ref = O.indice_conv_backward(x.numpy(), w.numpy().reshape(3, 3, 3, cin, cout), gy.numpy(), p_np, n_np)
            dw_ref = np.asarray(dw_ref).reshape(27, cin, cout)
            errs, outs = {}, {}
            for mode in ('f32', 'f32x6'):
                spconv.set_conv_precision(mode)
                y = spconv._gather_gemm(x.to(DEV), rb.out2in, m, w.to(DEV), False, cout, rb)
                dx = spconv._gather_gemm(gy.to(DEV), rb.in2out, n, w.to(DEV), True, cin, rb)
                # the filter gradient (csrc/spconv_os.hip sp_wgrad_os_x6_k beside sp_wgrad_os_k): same bar
                dw = spconv._wgrad(x.to(DEV), gy.to(DEV), rb, pairs, 0, (27, cin, cout))
                outs[mode] = (y, dw)
                errs[mode] = (np.abs(y.cpu().numpy() - y_ref).max(), np.abs(dx.cpu().numpy() - dx_ref).max(),
                              np.abs(dw.cpu().numpy() - dw_ref).max())
            sy, sx, sw = max(1.0, np.abs(y_ref).max()), max(1.0, np.abs(dx_ref).max()), max(1.0, np.abs(dw_ref).max())
            assert errs['f32x6'][0] <= max(2.0 * errs['f32'][0], 2e-7 * sy), (errs, sy)
            assert errs['f32x6'][1] <= max(2.0 * errs['f32'][1], 2e-7 * sx), (errs, sx)
            assert errs['f32x6'][2] <= max(2.0 * errs['f32'][2], 2e-7 * sw), (errs, sw)
            assert errs['f32x6'][0] <= 5e-6 * sy and errs['f32x6'][1] <= 5e-6 * sx and errs['f32x6'][2] <= 5e-6 * sw
            assert not torch.equal(outs['f32'][0], outs['f32x6'][0]) and not torch.equal(outs['f32'][1], outs['f32x6'][1])
    finally:
        spconv.set_conv_precision(spconv.DEFAULT_CONV_PRECISION)


def test_inverse_conv_matches_oracle_and_modules_chain():
    """SubMConv3d -> SparseConv3d (stride 2, indice_key) -> SubMConv3d -> SparseInverseConv3d back to the input voxels
    (the down / up pattern of middle_encoders/sparse_unet.py), forward and all gradients against the oracle."""
    from oracle import spconv_oracle as O
    from sst_amd import spconv
    rng = np.random.default_rng(4)
    batch, shape, n = 2, [9, 28, 30], 2200
    ind = _cloud(rng, n, batch, shape)
    torch.manual_seed(0)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(6, 16, 3, padding=1, bias=False, indice_key='subm1'),
        torch.nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=True, indice_key='down1'),
        spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key='subm2'),
        spconv.SparseInverseConv3d(32, 8, 3, indice_key='down1', bias=False)).to(DEV)
    x = torch.randn(n, 6)
    xa = x.to(DEV).requires_grad_(True)
    t = spconv.SparseConvTensor(xa, torch.from_numpy(ind).to(DEV), shape, batch)
    out = net(t)
    assert out.features.shape == (n, 8) and torch.equal(out.indices.cpu(), torch.from_numpy(ind))
    assert list(out.spatial_shape) == shape
    gy = torch.randn(n, 8)
    out.features.backward(gy.to(DEV))
    # oracle chain in float64
    w = [p.detach().cpu().numpy().astype(np.float64) for p in (net[0].weight, net[2].weight, net[3].weight, net[4].weight)]
    bias = net[2].bias.detach().cpu().numpy().astype(np.float64)
    _, p1, n1, _ = O.indice_pairs(ind, batch, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    o2, p2, n2, shape2 = O.indice_pairs(ind, batch, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3)
    _, p3, n3, _ = O.indice_pairs(o2, batch, shape2, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    h1 = O.indice_conv(x.numpy(), w[0], p1, n1, n)
    a1 = np.maximum(h1, 0)
    h2 = O.indice_conv(a1, w[1], p2, n2, len(o2)) + bias
    h3 = O.indice_conv(h2, w[2], p3, n3, len(o2))
    y = O.indice_conv(h3, w[3], p2, n2, n, inverse=True)
    assert np.abs(out.features.detach().cpu().numpy() - y).max() < 2e-4 * max(1.0, np.abs(y).max())
    d3, dw3 = O.indice_conv_backward(h3, w[3], gy.numpy(), p2, n2, inverse=True)
    d2, dw2 = O.indice_conv_backward(h2, w[2], d3, p3, n3)
    db = d2.sum(0)
    d1, dw1 = O.indice_conv_backward(a1, w[1], d2, p2, n2)
    d1 = d1 * (h1 > 0)
    dx, dw0 = O.indice_conv_backward(x.numpy(), w[0], d1, p1, n1)
    for got, want in ((xa.grad, dx), (net[0].weight.grad, dw0), (net[2].weight.grad, dw1), (net[2].bias.grad, db),
                      (net[3].weight.grad, dw2), (net[4].weight.grad, dw3)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - want).max()
        assert err < 5e-4 * max(1.0, np.abs(want).max()), err
    # the rulebook of 'down1' is shared by the inverse convolution, 'subm1' is reused by a second SubM with the key
    assert set(t.indice_dict) == {'subm1', 'down1', 'subm2'}


def test_spconv_edge_cases(monkeypatch):
    from sst_amd import spconv
    monkeypatch.setenv('SST_AMD_DEBUG', '1')     # the duplicate-voxel check costs a host read-back: debug mode only
    shape, batch = [5, 8, 8], 1
    conv = spconv.SubMConv3d(4, 8, 3, padding=1, indice_key='k').to(DEV)
    empty = spconv.SparseConvTensor(torch.zeros(0, 4, device=DEV), torch.zeros(0, 4, dtype=torch.int32, device=DEV),
                                    shape, batch)
    assert conv(empty).features.shape == (0, 8)
    one = spconv.SparseConvTensor(torch.ones(1, 4, device=DEV), torch.tensor([[0, 2, 3, 4]], dtype=torch.int32, device=DEV),
                                  shape, batch)
    y = conv(one)
    want = torch.ones(1, 4, device=DEV) @ conv.weight[1, 1, 1] + conv.bias
    assert torch.allclose(y.features, want, atol=1e-5)
    down = spconv.SparseConv3d(4, 8, 3, stride=2, padding=1, bias=False).to(DEV)
    z = down(one)
    assert z.indices.tolist() == [[0, 1, 1, 2], [0, 1, 2, 2]] and list(z.spatial_shape) == [3, 4, 4]
    with pytest.raises(RuntimeError):
        spconv.get_indice_pairs(torch.tensor([[0, 1, 1, 1], [0, 1, 1, 1]], dtype=torch.int32, device=DEV), 1, shape, 3,
                                subm=True)   # duplicate voxel
    with pytest.raises(RuntimeError):
        spconv.get_indice_pairs(torch.zeros(3, 4, dtype=torch.int32), 1, shape, 3, subm=True)   # CPU tensor
    d = one.dense()
    assert d.shape == (1, 4, 5, 8, 8) and float(d.sum()) == 4.0
    # pair lists that did not come from get_indice_pairs (no attached maps): rebuilt from the lists
    ind = torch.tensor([[0, 1, 1, 1], [0, 1, 1, 2], [0, 2, 2, 2]], dtype=torch.int32, device=DEV)
    outids, pairs, num = spconv.get_indice_pairs(ind, 1, shape, 3, subm=True)
    x = torch.randn(3, 4, device=DEV)
    a = spconv.indice_conv(x, conv.weight, pairs, num, 3, False, True)
    b = spconv.indice_conv(x, conv.weight, pairs.clone(), num, 3, False, True)
    assert torch.equal(a, b)


def test_spconv_full_size_properties():
    """FSD-scale SubM layer (150 k voxels, 64 -> 64): linearity, agreement of the gathered GEMM with the reference's
    per-offset gather / mm / index_add formulation in torch, determinism, rulebook symmetry."""
    from sst_amd import spconv
    rng = np.random.default_rng(8)
    batch, shape, n = 2, [41, 400, 400], 150000
    base = _cloud(rng, n // 4, batch, [shape[0], shape[1] // 2, shape[2] // 2])
    ind = np.unique(np.concatenate([base * [1, 1, 2, 2] + [0, 0, dy, dx] for dy in (0, 1) for dx in (0, 1)]), axis=0)
    ind = ind[rng.permutation(len(ind))].astype(np.int32)
    n = len(ind)
    outids, pairs, num = spconv.get_indice_pairs(torch.from_numpy(ind).to(DEV), batch, shape, 3, subm=True)
    rb = pairs._sst_rulebook
    assert int(num[13]) == n and torch.equal(rb.in2out[13], torch.arange(n, dtype=torch.int32, device=DEV))
    assert torch.equal(num, num.flip(0))                      # subm pairs come in mirrored offsets
    x = torch.randn(n, 64, device=DEV)
    w = torch.randn(3, 3, 3, 64, 64, device=DEV) * 0.05
    y = spconv.indice_conv(x, w, pairs, num, n, False, True)
    assert torch.equal(y, spconv.indice_conv(x, w, pairs, num, n, False, True))
    ref = torch.zeros(n, 64, device=DEV, dtype=torch.float64)
    w3 = w.view(27, 64, 64).double()
    for k in range(27):
        c = int(num[k])
        src, dst = pairs[k, 0, :c].long(), pairs[k, 1, :c].long()
        ref.index_add_(0, dst, x[src].double() @ w3[k])
    assert float((y.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    y2 = spconv.indice_conv(2.5 * x, w, pairs, num, n, False, True)
    assert float((y2 - 2.5 * y).abs().max()) < 1e-4 * float(y.abs().max())


@pytest.mark.parametrize('tag', ['down3s2', 'down2s2', 'subm3'])
def test_sparse_maxpool_matches_reference_golden(tag):
    """indice_maxpool forward / backward (bit-exact: max and exact-equality selections, sums of at most K terms in the
    reference's offset order differ only by association -> 1e-6) against the reference's CPU functors (golden)"""
    from sst_amd import spconv
    g = load_golden('spconv.npz')
    ind, batch, shape, ks, st, pd, dl, subm, tr = spconv_case(g, tag)
    outids, pairs, num, rb = _rulebook(ind, batch, shape, ks, st, pd, dl, subm, tr)
    pos = {tuple(c): i for i, c in enumerate(g[f'out::{tag}::outids'].tolist())}
    perm = np.array([pos[tuple(c)] for c in outids.cpu().tolist()])
    x = torch.from_numpy(g[f'in::{tag}::pool_features']).to(DEV).requires_grad_(True)
    y = spconv.indice_maxpool_fn(x, pairs, num, len(outids))
    np.testing.assert_array_equal(y.detach().cpu().numpy(), g[f'out::{tag}::pooled'][perm])
    y.backward(torch.from_numpy(g[f'in::{tag}::pool_grad_out'][perm]).to(DEV))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'out::{tag}::pool_grad_in'], atol=1e-6)


def test_sparse_maxpool_module():
    from sst_amd import spconv
    rng = np.random.default_rng(12)
    ind = _cloud(rng, 800, 2, [8, 20, 20])
    t = spconv.SparseConvTensor(torch.randn(800, 5, device=DEV), torch.from_numpy(ind).to(DEV), [8, 20, 20], 2)
    out = spconv.SparseMaxPool3d(3, stride=2, padding=1)(t)
    assert list(out.spatial_shape) == [4, 10, 10] and out.features.shape[1] == 5 and (out.features >= 0).all()
    dense_in = t.dense().clamp(min=0)
    want = torch.nn.functional.max_pool3d(dense_in, 3, 2, 1)       # zeros where nothing is active, like the sparse op
    got = out.dense()
    assert torch.equal(got, want * (got != 0)) or torch.allclose(got[got != 0], want[got != 0])


@pytest.mark.gpu
@pytest.mark.parametrize('subm', [True, False])
def test_grid_rulebook_ignores_rows_outside_the_grid(subm):
    """the dense-grid builders index batch x shape cell arrays with the input coordinates - a sample index
    >= batch_size, a negative coordinate or one >= spatial_shape must not write out of bounds.  Such rows take no part: the
    valid rows get exactly the rulebook they get alone, the bad rows no pair."""
    from sst_amd import spconv
    rng = np.random.RandomState(5)
    shape, batch = [8, 12, 12], 2
    cells = rng.choice(batch * 8 * 12 * 12, 300, replace=False)
    good = np.stack([cells // 1152, cells // 144 % 8, cells // 12 % 12, cells % 12], 1).astype(np.int32)
    bad = np.array([[2, 1, 1, 1], [0, 8, 0, 0], [1, 0, 12, 3], [0, 0, 0, -1], [-1, 2, 2, 2], [0, 0, 0, 4000]], dtype=np.int32)
    mixed = np.concatenate([good[:150], bad[:3], good[150:], bad[3:]])
    pos_good = np.concatenate([np.arange(150), np.arange(153, 303)])
    kw = dict(ksize=3, stride=1 if subm else 2, padding=1, subm=subm)
    out_a, pairs_a, num_a = spconv.get_indice_pairs(torch.from_numpy(good).to(DEV), batch, shape, **kw)
    out_b, pairs_b, num_b = spconv.get_indice_pairs(torch.from_numpy(mixed).to(DEV), batch, shape, **kw)
    torch.cuda.synchronize()
    assert torch.equal(num_a, num_b)
    if not subm:
        assert torch.equal(out_a, out_b)
    remap = torch.from_numpy(pos_good).to(DEV)
    for k in range(27):
        c = int(num_a[k])
        a_in, b_in = remap[pairs_a[k, 0, :c].long()], pairs_b[k, 0, :c].long()
        assert torch.equal(a_in, b_in)
        if subm:      # output rows = input rows: renumbered the same way
            assert torch.equal(remap[pairs_a[k, 1, :c].long()], pairs_b[k, 1, :c].long())
        else:
            assert torch.equal(pairs_a[k, 1, :c], pairs_b[k, 1, :c])


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout', [(64, 64), (128, 128)])
def test_sparse_filter_gradient_bit_reproducible(cin, cout):
    """the 300-launch loop over the second cross-workgroup hand-over of the library - the chunk partials of the
    output-stationary filter gradient and their reduction (sp_wgrad_os_k / sp_wgrad_os_reduce_k, csrc/spconv_os.hip) - on a
    LiDAR-like voxel set large enough for several chunks per offset: forward, data gradient and filter gradient, same bits"""
    from sst_amd import spconv
    rng = np.random.RandomState(7)
    shape, batch = [16, 200, 200], 2
    cells = rng.choice(batch * 16 * 200 * 200, 60000, replace=False)
    cells.sort()
    ind = np.stack([cells // (16 * 200 * 200), cells // 40000 % 16, cells // 200 % 200, cells % 200], 1).astype(np.int32)
    # concentrate the voxels near the ground (z < 4) so that the submanifold rulebook has populated offsets
    ind[:, 1] = ind[:, 1] % 4
    ind = np.unique(ind, axis=0)
    outids, pairs, num = spconv.get_indice_pairs(torch.from_numpy(ind).to(DEV), batch, shape, 3, subm=True)
    gen = torch.Generator().manual_seed(cin)
    x = torch.randn(len(ind), cin, generator=gen).to(DEV)
    w = (torch.randn(3, 3, 3, cin, cout, generator=gen) * 0.2).to(DEV)
    gy = torch.randn(len(ind), cout, generator=gen).to(DEV)
    ref = None
    for _ in range(300):
        xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = spconv.indice_subm_conv(xa, wa, pairs, num, len(ind))
        y.backward(gy)
        cur = (y.detach(), xa.grad, wa.grad)
        if ref is None:
            ref = tuple(t.clone() for t in cur)
            assert int(num.sum()) > 200000        # several 2048-pair chunks per offset
        assert all(torch.equal(a, b) for a, b in zip(ref, cur))


@pytest.mark.parametrize('cin,cout,n', [(256, 256, 1866), (128, 128, 6542), (64, 128, 700), (256, 128, 3000)])
def test_thin_levels_deal_their_offsets_out(cin, cout, n):
    """csrc/spconv_os_x6.hip on a THIN level (the two deepest levels of FSD's U-Net: 1 866 / 6 542 rows): with room for partial
    tiles in its workspace the call splits the 27 offsets over up to 8 workgroups per (row tile, column group) and adds the
    partials in a fixed order - same values as the unsplit call up to the association of the per-offset sums, the same bits
    every launch, bias included."""
    from sst_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n)
    batch, shape = 1, [8, 48, 48]
    ind = _cloud(rng, n, batch, shape)
    outids, pairs, num, rb = _rulebook(ind, batch, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, False)
    m = len(outids)
    gen = torch.Generator().manual_seed(n)
    x = (torch.randn(n, cin, generator=gen) * 2).to(DEV)
    w = (torch.randn(27, cin, cout, generator=gen) * 0.1).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    outs = []
    for rows_ws in (False, True, True):
        y = torch.empty(m, cout, device=DEV)
        nbytes = (lib.sst_spconv_conv_os_f32x6_workspace_bytes_rows(27, cin, cout, m) if rows_ws
                  else lib.sst_spconv_conv_os_f32x6_workspace_bytes(27, cin, cout))
        ws = _lib.workspace(nbytes, x.device)
        rc = lib.sst_spconv_conv_os_rows_f32x6(_lib.ptr(x), cin, _lib.ptr(rb.out2in), m, 27, _lib.ptr(w), cin, cout, 0,
                                               _lib.ptr(bias), _lib.ptr(y), cout, 0, None, _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr())
        assert rc == 0
        outs.append(y)
    assert lib.sst_spconv_conv_os_f32x6_workspace_bytes_rows(27, cin, cout, m) > \
        lib.sst_spconv_conv_os_f32x6_workspace_bytes(27, cin, cout) + m * cout * 4       # the split is on at this size
    scale = float(outs[0].abs().max())
    assert float((outs[1] - outs[0]).abs().max()) <= 2e-6 * scale
    assert torch.equal(outs[1], outs[2])
