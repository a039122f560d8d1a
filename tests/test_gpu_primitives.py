"""GPU: device primitives (scan, radix sort, unique, in-group rank) against numpy, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 2047, 2048, 2049, 100003, 1 << 20])
def test_exclusive_scan(n):
    from sst_amd import kernels as K
    rng = np.random.default_rng(n)
    x = rng.integers(0, 7, size=n).astype(np.int32)
    out, total = K.exclusive_scan_i32(torch.from_numpy(x).to(_dev()))
    ref = np.concatenate([[0], np.cumsum(x)[:-1]]).astype(np.int32) if n else x
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    assert int(total.item()) == int(x.sum())


@pytest.mark.parametrize('n,bits', [(1, 1), (777, 8), (2048, 9), (5000, 16), (100003, 20), (300000, 33), (4097, 63)])
def test_radix_sort_pairs_stable(n, bits):
    from sst_amd import kernels as K
    rng = np.random.default_rng(n + bits)
    hi = (1 << bits) - 1
    # many duplicates to exercise stability
    keys = rng.integers(0, min(hi, max(2, n // 3)) + 1, size=n, dtype=np.int64)
    if bits > 20:
        keys = keys | (rng.integers(0, 2, size=n, dtype=np.int64) << (bits - 1))
    sk, perm = K.sort_pairs_u64(torch.from_numpy(keys).to(_dev()), bits)
    order = np.argsort(keys, kind='stable')
    np.testing.assert_array_equal(sk.cpu().numpy(), keys[order])
    np.testing.assert_array_equal(perm.cpu().numpy().astype(np.int64), order)


@pytest.mark.parametrize('dtype', [torch.int32, torch.int64])
@pytest.mark.parametrize('n', [1, 5, 3000, 120000])
def test_unique_rows_matches_torch_unique(n, dtype):
    from sst_amd import kernels as K
    g = torch.Generator().manual_seed(n)
    coors = torch.stack([torch.randint(0, 2, (n,), generator=g), torch.randint(-3, 40, (n,), generator=g),
                         torch.randint(0, 25, (n,), generator=g), torch.randint(5, 9, (n,), generator=g)], 1).to(dtype)
    plan = K.unique_rows(coors.to(_dev()).contiguous())
    uniq, inv, cnt = torch.unique(coors, dim=0, return_inverse=True, return_counts=True)
    assert plan.m == uniq.size(0)
    np.testing.assert_array_equal(plan.inverse.cpu().numpy(), inv.numpy())
    np.testing.assert_array_equal(plan.counts().cpu().numpy(), cnt.numpy())
    np.testing.assert_array_equal(K.unpack_unique_rows(plan, dtype).cpu().numpy(), uniq.numpy())
    # CSR: perm groups rows by unique id, ascending row index inside a group
    perm = plan.perm.cpu().numpy().astype(np.int64)
    off = plan.offsets.cpu().numpy()[:plan.m + 1]
    order = np.lexsort((np.arange(n), inv.numpy()))
    np.testing.assert_array_equal(perm, order)
    assert off[0] == 0 and off[-1] == n


def test_unique_rows_invalid_group_sorts_first():
    from sst_amd import kernels as K
    coors = torch.tensor([[0, 1, 1], [-1, 3, 3], [0, 0, 5], [2, -1, 0], [0, 1, 1]], dtype=torch.int32)
    plan = K.unique_rows(coors.to(_dev()), invalid_if_negative=1)
    assert plan.m == 3
    assert plan.inverse.cpu().tolist() == [2, 0, 1, 0, 2]
    rows = K.unpack_unique_rows(plan, torch.int32).cpu().tolist()
    assert rows == [[-1, -1, -1], [0, 0, 5], [0, 1, 1]]


@pytest.mark.parametrize('n', [1, 1000, 90001])
def test_ingroup_rank(n):
    import sst_amd
    from oracle import sst_oracle
    rng = np.random.default_rng(n)
    ids = rng.integers(0, max(2, n // 50), size=n).astype(np.int64)
    out = sst_amd.get_inner_win_inds(torch.from_numpy(ids).to(_dev()))
    np.testing.assert_array_equal(out.cpu().numpy(), sst_oracle.ingroup_rank(ids))


def test_make_continuous_inds():
    import sst_amd
    ids = torch.tensor([70, 3, 3, 1000, 70, 5], dtype=torch.long)
    out = sst_amd.make_continuous_inds(ids.to(_dev()))
    assert out.cpu().tolist() == [2, 0, 0, 3, 2, 1]


@pytest.mark.parametrize('c', [128, 32, 64, 16, 3, 5, 256])
@pytest.mark.parametrize('n,k,first', [(18443, 1554, 0), (50000, 1554, 0), (4097, 60, 0), (5000, 333, 1), (100, 3, 0), (9, 1, 0)])
def test_long_group_reduce_equals_the_csr_walk(n, k, first, c, monkeypatch):
    """FSD-like groupings (Zipf cluster sizes, thousands of points in the largest): the tile kernel (seg_tiles_k: groups
    that cross tiles are merged by the last of their tiles to arrive) against the per-group CSR kernels.  MAX: values AND
    arg-max rows bit-identical (ties -> smallest row index), gradients through the recorded arg-max; repeated calls reuse the
    self-cleaning counters; SUM / MEAN against float64."""
    from sst_amd import kernels as K
    rng = np.random.default_rng(n + c)
    w = 1.0 / np.arange(1, k + 1) ** 1.1
    ids = rng.choice(k, size=n, p=w / w.sum())
    ids[:k] = np.arange(k)                                  # every group has a point
    coors = torch.from_numpy(np.stack([np.zeros(n, np.int64), ids // 7, ids % 7], 1)).to(_dev())
    plan = K.unique_rows(coors)
    assert plan.m == k and n >= 8 * (k - first)
    feats = torch.from_numpy(rng.integers(-3, 4, size=(n, c)).astype(np.float32)).to(_dev())   # many exact ties
    feats[rng.integers(0, n, 50)] = 0.0
    feats[rng.integers(0, n, 50)] = -0.0
    outs = []
    for flag in ('1', '0', '1'):
        monkeypatch.setenv('SST_SEG_LONG', flag)
        x = feats.clone().requires_grad_(True)
        y = K.segment_reduce(x, plan, 'max', first=first)
        y.backward(torch.ones_like(y) * torch.arange(1, c + 1, device=_dev()))
        outs.append((y.detach(), x.grad.clone()))
    (y_new, g_new), (y_old, g_old), (y_again, g_again) = outs
    assert torch.equal(y_new, y_old) and torch.equal(g_new, g_old)          # same values, same arg-max rows
    assert torch.equal(y_new, y_again) and torch.equal(g_new, g_again)      # the counters cleaned themselves
    inv = plan.inverse.long()[:, None].expand(n, c)
    ref = torch.full((k, c), float('-inf'), device=_dev()).scatter_reduce(0, inv, feats, reduce='amax')
    assert torch.equal(y_new, ref[first:])
    monkeypatch.setenv('SST_SEG_LONG', '1')
    real = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).to(_dev())
    want = torch.zeros((k, c), dtype=torch.float64, device=_dev()).index_add_(0, plan.inverse.long(), real.double())
    got_sum = K.segment_reduce(real, plan, 'sum', first=first)
    got_mean = K.segment_reduce(real, plan, 'mean', first=first)
    cnt = torch.bincount(plan.inverse.long(), minlength=k).double()[:, None]
    assert float((got_sum.double() - want[first:]).abs().max()) < 1e-3
    assert float((got_mean.double() - (want / cnt)[first:]).abs().max()) < 1e-5
    assert torch.equal(got_sum, K.segment_reduce(real, plan, 'sum', first=first))     # deterministic


@pytest.mark.parametrize('n,k,c', [(50000, 1554, 256), (100000, 3000, 128), (18443, 1554, 3)])
def test_long_group_reduce_gives_the_same_answer_every_launch(n, k, c):
    """the hand-over of the partial records between workgroups on different XCDs (seg_tiles_k): 300 launches each of MAX and
    SUM against one reference - a record that travels without the right scope / ordering shows up as a rare wrong group
    (an earlier version of the kernel failed 1-8 % of these launches)."""
    from sst_amd import kernels as K
    rng = np.random.default_rng(n + c)
    w = 1.0 / np.arange(1, k + 1) ** 1.1
    ids = rng.choice(k, size=n, p=w / w.sum())
    ids[:k] = np.arange(k)
    coors = torch.from_numpy(np.stack([np.zeros(n, np.int64), ids // 7, ids % 7], 1)).to(_dev())
    plan = K.unique_rows(coors)
    feats = torch.from_numpy(rng.integers(-3, 4, size=(n, c)).astype(np.float32)).to(_dev())
    inv = plan.inverse.long()[:, None].expand(n, c)
    ref = torch.full((k, c), float('-inf'), device=_dev()).scatter_reduce(0, inv, feats, reduce='amax')
    ref_sum = torch.zeros((k, c), dtype=torch.float64, device=_dev()).index_add_(0, plan.inverse.long(), feats.double())
    bad = 0
    for _ in range(300):
        bad += int(not torch.equal(K.segment_reduce(feats, plan, 'max'), ref))
        bad += int(not torch.equal(K.segment_reduce(feats, plan, 'sum').double(), ref_sum))   # small integers: exact
    assert bad == 0



@pytest.mark.parametrize('c', [128, 64, 3, 11])
@pytest.mark.parametrize('subset', [False, True])
def test_voxel_grouping_with_a_few_huge_voxels(c, subset, monkeypatch):
    """A real sweep's voxel grouping: 1-3 points in most voxels, thousands in the voxels next to the sensor (n < 8 m, so
    the long-group kernels chosen by the AVERAGE never run).  The work-list form (long groups listed by the element kernel,
    one workgroup each) against the serial walk: MAX bit-identical incl. arg-max rows, SUM / MEAN against float64, every
    launch the same, the list cleans itself; `subset`: through group_index + a device-side row limit (the batched
    DynamicScatter form)."""
    from sst_amd import kernels as K
    rng = np.random.default_rng(c + subset)
    k, n_short, huge = 20000, 30000, [5000, 3000, 2000, 700, 300, 100, 40, 33]
    ids = np.concatenate([np.arange(k), rng.integers(0, k, n_short - k)] + [np.full(h, 17 + 1000 * i) for i, h in enumerate(huge)])
    rng.shuffle(ids)
    n = ids.size
    assert n < 8 * k
    coors = torch.from_numpy(np.stack([np.zeros(n, np.int64), ids // 200, ids % 200], 1)).to(_dev())
    plan = K.unique_rows(coors)
    assert plan.m == k
    feats = torch.from_numpy(rng.integers(-3, 4, size=(n, c)).astype(np.float32)).to(_dev())
    real = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).to(_dev())
    if subset:
        gidx = torch.arange(1, k, dtype=torch.int32, device=_dev())        # all but the first group
        inverse = plan.inverse - 1
        m_limit = torch.tensor([k - 5], dtype=torch.int32, device=_dev())  # the last 4 rows are not written
        kw = dict(group_index=gidx, inverse=inverse, m_limit=m_limit)
        rows = slice(1, k - 4)
        nrows = k - 5
    else:
        kw = {}
        rows = slice(0, k)
        nrows = k
    res = {}
    for flag in ('1', '0', '1'):
        monkeypatch.setenv('SST_SEG_WORK', flag)
        x = feats.clone().requires_grad_(True)
        y = K.segment_reduce(x, plan, 'max', **kw)
        y[:nrows].backward(torch.ones_like(y[:nrows]) * torch.arange(1, c + 1, device=_dev()))
        res.setdefault(flag, []).append((y.detach()[:nrows].clone(), x.grad.clone()))
    (y_new, g_new), (y_again, g_again) = res['1']
    (y_old, g_old), = res['0']
    assert torch.equal(y_new, y_old) and torch.equal(g_new, g_old)
    assert torch.equal(y_new, y_again) and torch.equal(g_new, g_again)
    inv = plan.inverse.long()[:, None].expand(n, c)
    ref = torch.full((k, c), float('-inf'), device=_dev()).scatter_reduce(0, inv, feats, reduce='amax')
    assert torch.equal(y_new, ref[rows])
    monkeypatch.setenv('SST_SEG_WORK', '1')
    want = torch.zeros((k, c), dtype=torch.float64, device=_dev()).index_add_(0, plan.inverse.long(), real.double())
    cnt = torch.bincount(plan.inverse.long(), minlength=k).double()[:, None]
    got_sum = K.segment_reduce(real, plan, 'sum', **kw)[:nrows]
    got_mean = K.segment_reduce(real, plan, 'mean', **kw)[:nrows]
    assert float((got_sum.double() - want[rows]).abs().max()) < 2e-3
    assert float((got_mean.double() - (want / cnt)[rows]).abs().max()) < 1e-5
    for _ in range(20):
        assert torch.equal(got_sum, K.segment_reduce(real, plan, 'sum', **kw)[:nrows])
    val, arg = K.segment_argmax(feats, plan)
    assert torch.equal(val, ref)
    assert torch.equal(feats.gather(0, arg.long()), ref)


@pytest.mark.parametrize('n,cap', [(1, 5), (600, 100), (1521, 100), (5000, 37), (70000, 100), (3000, 511), (900, 700)])
def test_window_launch_order_equals_the_stable_sort(n, cap):
    """WindowPlan.order (csrc/window.hip window_order_k: one launch) == torch.sort(sizes, stable=True)[1]; windows of 512 tokens
    and more take the library sort"""
    from sst_amd import kernels as K
    g = torch.Generator().manual_seed(n + cap)
    sizes = torch.randint(0, cap + 1, (n,), generator=g)
    sizes[torch.randint(0, n, (max(1, n // 7),), generator=g)] = cap          # many ties at the cap, as after the drop
    winoff = torch.cat([torch.zeros(1, dtype=torch.int64), sizes.cumsum(0)]).to(torch.int32).to(_dev())
    total = int(sizes.sum())
    plan = K.WindowPlan(torch.arange(max(total, 1), dtype=torch.int32, device=_dev()), winoff, n, total, cap)
    old = K.WINDOW_ORDER_MIN
    K.WINDOW_ORDER_MIN = 1
    try:
        order = plan.order
    finally:
        K.WINDOW_ORDER_MIN = old
    ref = torch.sort(sizes, stable=True)[1].to(torch.int32)
    assert order.dtype == torch.int32 and torch.equal(order.cpu(), ref)


def test_add_table_rows_equals_index_select_plus_add():
    from sst_amd import kernels as K
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5003, 128, generator=g).to(_dev()).requires_grad_(True)
    table = torch.randn(144, 128, generator=g).to(_dev())
    idx = torch.randint(0, 144, (5003,), generator=g, dtype=torch.int32).to(_dev())
    y = K.add_table_rows(x, table, idx)
    assert torch.equal(y, x + table.index_select(0, idx.long()))
    w = torch.randn(y.shape, generator=g).to(_dev())
    (y * w).sum().backward()
    assert torch.equal(x.grad, w)
