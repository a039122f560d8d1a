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
