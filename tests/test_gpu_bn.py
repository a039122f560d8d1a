"""BatchNorm1d (+ReLU) kernels (csrc/bn.hip) against torch's own batch_norm in float64, and the naiveSyncBN
semantics of the reference (mmdet3d/ops/norm.py:28-86) with two ranks sharing the GPU (gloo rendezvous)."""
import os
import socket

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3  # north-star feature tolerance; observed errors are ~1e-6


def _dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('n,c', [(1, 4), (7, 36), (1000, 64), (5001, 128), (116000, 128), (30000, 256), (257, 1024)])
@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('training', [True, False])
def test_batch_norm_act_matches_torch(n, c, relu, training):
    from sst_amd.norm import BatchNorm1d, batch_norm_act
    if n == 1 and training:
        pytest.skip('torch refuses batch statistics of a single row')
    torch.manual_seed(n + c)
    dev = _dev()
    x = (torch.randn(n, c, device=dev) * 3 + torch.linspace(-50, 50, c, device=dev)).requires_grad_(True)
    bn = BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).double()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(-1.0, 2.0, c))
        bn.bias.copy_(torch.linspace(-0.5, 0.5, c))
        bn.running_mean.copy_(torch.linspace(-40, 40, c))
        bn.running_var.copy_(torch.linspace(5, 12, c))
        ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bn.train(training)
    ref.train(training)
    y = batch_norm_act(bn, x, relu=relu)
    xr = x.detach().double().requires_grad_(True)
    pre = ref(xr)
    yr = F.relu(pre) if relu else pre
    gy = torch.randn(n, c, device=dev)
    y.backward(gy)
    yr.backward(gy.double())
    assert (y.double() - yr).abs().max().item() < TOL * 0.1
    scale = max(1.0, xr.grad.abs().max().item())
    # ReLU is discontinuous: outputs within rounding of 0 may take the other branch in fp32, leave them out
    safe = (pre.detach().abs() > 1e-4) if relu else torch.ones_like(yr, dtype=torch.bool)
    assert ((x.grad.double() - xr.grad).abs() * safe).max().item() < TOL * 0.1 * scale
    assert (~safe).float().mean().item() < 1e-2
    # an element that takes the other ReLU branch moves a column sum by up to |gy| * max(1, |xhat|)
    flip = ((gy.double().abs() * (~safe)).sum(0) * 6.0).max().item()
    for p, q in ((bn.weight, ref.weight), (bn.bias, ref.bias)):
        s = max(1.0, q.grad.abs().max().item())
        assert (p.grad.double() - q.grad).abs().max().item() < 1e-4 * s + flip
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, atol=1e-5, rtol=1e-5)
    assert torch.allclose(bn.running_var.double(), ref.running_var, atol=1e-5, rtol=1e-5)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize('n,c', [(7, 36), (5001, 64), (60000, 128)])
@pytest.mark.parametrize('training', [True, False])
def test_batch_norm_residual_relu_matches_torch(n, c, training):
    """relu(bn(x) + identity) in the batch-norm passes (the tail of a sparse basic block, sparse_block.py:127-139): output,
    the gradients of x, of the identity branch and of the affine parameters, and the bookkeeping, against torch in
    float64; then the block itself (SparseBasicBlock) with the fused passes against its reference order of operations."""
    from sst_amd.norm import BatchNorm1d, batch_norm_act
    torch.manual_seed(n + c)
    dev = _dev()
    x = (torch.randn(n, c, device=dev) * 2 + torch.linspace(-5, 5, c, device=dev)).requires_grad_(True)
    res = torch.randn(n, c, device=dev).requires_grad_(True)
    bn = BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).double()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(-1.0, 2.0, c))
        bn.bias.copy_(torch.linspace(-0.5, 0.5, c))
        ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bn.train(training)
    ref.train(training)
    y = batch_norm_act(bn, x, relu=True, residual=res)
    xr, rr = x.detach().double().requires_grad_(True), res.detach().double().requires_grad_(True)
    pre = ref(xr) + rr
    yr = F.relu(pre)
    gy = torch.randn(n, c, device=dev)
    y.backward(gy)
    yr.backward(gy.double())
    assert (y.double() - yr).abs().max().item() < TOL * 0.1
    safe = pre.detach().abs() > 1e-4
    assert (~safe).float().mean().item() < 1e-2
    scale = max(1.0, xr.grad.abs().max().item())
    assert ((x.grad.double() - xr.grad).abs() * safe).max().item() < TOL * 0.1 * scale
    assert ((res.grad.double() - rr.grad).abs() * safe).max().item() < 1e-6
    flip = ((gy.double().abs() * (~safe)).sum(0) * 6.0).max().item()
    for p, q in ((bn.weight, ref.weight), (bn.bias, ref.bias)):
        assert (p.grad.double() - q.grad).abs().max().item() < 1e-4 * max(1.0, q.grad.abs().max().item()) + flip
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == (1 if training else 0)
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, atol=1e-5, rtol=1e-5)


def test_sparse_basic_block_fused_tail_equals_the_composed_one():
    from sst_amd import spconv
    from sst_amd.sparse_unet import SparseBasicBlock
    torch.manual_seed(5)
    dev = _dev()
    n, c = 3000, 32
    coords = torch.unique(torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.randint(0, 24, (n, 3), dtype=torch.int32)], 1), dim=0)
    feats = torch.randn(coords.size(0), c)
    blk = SparseBasicBlock(c, c, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                           conv_cfg=dict(type='SubMConv3d', indice_key='k')).to(dev).train()
    outs = []
    for fused in (True, False):
        blk.zero_grad()
        f = feats.to(dev).requires_grad_(True)
        t = spconv.SparseConvTensor(f, coords.to(dev), [24, 24, 24], 1)
        if not fused:
            blk.relu = torch.nn.ReLU(inplace=False)
            blk.relu.__class__ = type('ReLUComposed', (torch.nn.ReLU,), {})   # not `nn.ReLU` itself: the composed order runs
        y = blk(t).features
        y.backward(torch.ones_like(y) * 0.01)
        outs.append((y.detach(), f.grad.clone(), blk.conv1.weight.grad.clone(), blk.norm2.weight.grad.clone()))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


def test_batch_norm_strided_rows_and_module_forward():
    from sst_amd.norm import BatchNorm1d, build_norm_layer
    dev = _dev()
    torch.manual_seed(3)
    big = torch.randn(4000, 192, device=dev)
    x = big[:, 64:128]  # row stride 192, 16-byte aligned
    name, bn = build_norm_layer(dict(type='BN1d', eps=1e-3, momentum=0.01), 64)
    assert name == 'bn' and isinstance(bn, BatchNorm1d)
    bn = bn.to(dev).train()
    ref = torch.nn.BatchNorm1d(64, eps=1e-3, momentum=0.01).to(dev).train()
    assert (bn(x) - ref(x.contiguous())).abs().max().item() < 1e-5
    # 3-D input: the module's own (library) forward
    x3 = torch.randn(8, 64, 5, device=dev)
    assert bn(x3).shape == x3.shape


def test_bn_entry_points_reject_bad_arguments():
    from sst_amd import _lib
    lib = _lib.load()
    x = torch.randn(16, 6, device=_dev())
    out = torch.empty(2, 6, device=_dev())
    ws = _lib.workspace(lib.sst_bn_workspace_bytes(16, 6), x.device)
    # c % 4 != 0 -> SST_ERR_UNSUPPORTED (negative), never a silent fallback
    rc = lib.sst_bn_stats_f32(_lib.ptr(x), 16, 6, 6, _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(ws), _lib.stream_ptr())
    assert rc < 0


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sync_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sst_amd.norm import NaiveSyncBatchNorm1d, batch_norm_act
        dev = torch.device('cuda:0')
        torch.manual_seed(0)
        full = torch.randn(3000, 64) * 2 + 1
        gfull = torch.randn(3000, 64)
        sizes = [1800, 1200]
        start = sum(sizes[:rank])
        x = full[start:start + sizes[rank]].to(dev).requires_grad_(True)
        bn = NaiveSyncBatchNorm1d(64, eps=1e-3, momentum=0.01).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 64))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, 64))
        y = batch_norm_act(bn, x, relu=True)
        y.backward(gfull[start:start + sizes[rank]].to(dev))
        ret[rank] = dict(y=y.detach().cpu(), gx=x.grad.cpu(), gw=bn.weight.grad.cpu(), gb=bn.bias.grad.cpu(),
                         rm=bn.running_mean.cpu(), rv=bn.running_var.cpu())
    finally:
        dist.destroy_process_group()


def test_naive_sync_bn_two_ranks_on_one_gpu():
    """Statistics = plain average over ranks of per-rank mean / mean-of-squares; the gradient flows through the
    averaged statistics of BOTH ranks (autograd of the reference formulation, evaluated here on CPU in fp64)."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    torch.manual_seed(0)
    full = (torch.randn(3000, 64) * 2 + 1).double()
    gfull = torch.randn(3000, 64).double()
    parts = [full[:1800].clone().requires_grad_(True), full[1800:].clone().requires_grad_(True)]
    w = torch.linspace(0.5, 1.5, 64).double().requires_grad_(True)
    b = torch.linspace(-0.2, 0.2, 64).double().requires_grad_(True)
    mean = sum(p.mean(0) for p in parts) / 2
    meansqr = sum((p * p).mean(0) for p in parts) / 2
    var = meansqr - mean * mean
    outs = [F.relu((p - mean) * torch.rsqrt(var + 1e-3) * w + b) for p in parts]
    # per-rank parameter gradients are local sums (DDP averages them afterwards)
    for r, (o, g) in enumerate(zip(outs, (gfull[:1800], gfull[1800:]))):
        gw, gb = torch.autograd.grad(o, (w, b), g, retain_graph=True)
        assert (ret[r]['gw'].double() - gw).abs().max().item() < 1e-3 * max(1.0, gw.abs().max().item())
        assert (ret[r]['gb'].double() - gb).abs().max().item() < 1e-3 * max(1.0, gb.abs().max().item())
    loss = (outs[0] * gfull[:1800]).sum() + (outs[1] * gfull[1800:]).sum()
    gx = torch.autograd.grad(loss, parts)
    for r in range(2):
        assert (ret[r]['y'].double() - outs[r].detach()).abs().max().item() < 1e-4
        assert (ret[r]['gx'].double() - gx[r]).abs().max().item() < 1e-4
        assert torch.allclose(ret[r]['rm'].double(), 0.01 * mean.detach(), atol=1e-6)
        assert torch.allclose(ret[r]['rv'].double(), 0.99 + 0.01 * var.detach(), atol=1e-5)
