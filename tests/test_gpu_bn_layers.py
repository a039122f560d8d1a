"""GPU: encoder layers with batch norm (layer_cfg use_bn=True: sst_basic_block_v2.py:92-99, configs/fsd/fsd_waymoD1_1x_sst_encoder.py:70)
on the fused chain - the node of sst_amd/sst_basic_block.py FusedEncoderLayerFn with its batch-norm tail - against the module-by-
module path (whose pieces are pinned by the reference golden sst_block_bn_cosine.npz in test_gpu_sra.py, both paths), in training
and in evaluation mode, and the batch-norm row passes under naiveSyncBN's two collectives (two gloo ranks on one device) against
one float64 batch norm over the joint rows."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import DROP_TEST, DROP_TRAIN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(layer_cfg, blocks=2):
    import sst_amd
    torch.manual_seed(3)
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128] * blocks, nhead=[8] * blocks, num_blocks=blocks,
                                      dim_feedforward=[256] * blocks, output_shape=[468, 468], num_attached_conv=0, to_bev=False,
                                      debug=False, layer_cfg=layer_cfg))
    return net.to(DEV)


def _frame(n=6000, batch=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    coors = torch.unique(torch.stack([torch.randint(0, batch, (n,), generator=g), torch.zeros(n, dtype=torch.long),
                                      torch.randint(0, 468, (n,), generator=g), torch.randint(0, 468, (n,), generator=g)], 1), dim=0)
    feats = torch.randn(coors.size(0), 128, generator=g)
    return feats.to(DEV), coors.int().to(DEV)


def _run(net, fused, train, feats, coors, grad_out):
    import sst_amd
    net.set_fused(fused)
    net.train(train)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False, mute=True)
    layer.eval()
    x = feats.clone().requires_grad_(True)
    for p in net.parameters():
        p.grad = None
    out = net(layer(x, coors, 2))[0]['voxel_feats']
    (out * grad_out).sum().backward()
    return out.detach(), x.grad.detach(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('layer_cfg', [dict(use_bn=True), dict(use_bn=True, cosine=True, tau_min=0.01)])
@pytest.mark.parametrize('train', [True, False])
def test_batch_norm_layers_run_on_the_fused_chain(layer_cfg, train, monkeypatch):
    from sst_amd import sst_basic_block as B
    net = _net(layer_cfg)
    feats, coors = _frame()
    grad_out = torch.randn(feats.shape, device=DEV)
    calls = []
    orig = B.FusedEncoderLayerFn._tail_batch_norm
    monkeypatch.setattr(B.FusedEncoderLayerFn, '_tail_batch_norm', staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1]))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    out_f, gin_f, grads_f = _run(net, True, train, feats, coors, grad_out)
    assert len(calls) == 4, 'the four layers of the two blocks must take the batch-norm tail of the fused node'
    stats_f = {k: v.clone() for k, v in net.state_dict().items() if 'running' in k}
    net.load_state_dict(sd)
    out_m, gin_m, grads_m = _run(net, False, train, feats, coors, grad_out)
    assert len(calls) == 4
    assert float((out_f - out_m).abs().max()) < 2e-4
    assert float((gin_f - gin_m).abs().max()) < 2e-4 * max(1.0, float(gin_m.abs().max()))
    assert grads_f.keys() == grads_m.keys()
    for k in grads_m:
        scale = max(1.0, float(grads_m[k].abs().max()))
        assert float((grads_f[k] - grads_m[k]).abs().max()) < 5e-4 * scale, k
    for k, v in net.state_dict().items():        # the modules' bookkeeping (running statistics) moved the same way
        if 'running' in k:
            assert torch.allclose(v, stats_f[k], rtol=1e-5, atol=1e-6), k


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(seed, n=3000, c=128):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, c, generator=g) * 2.0 + 0.5, torch.randn(n, c, generator=g)


def _sync_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(DEV)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sst_amd import sst_basic_block as B
        from sst_amd.norm import NaiveSyncBatchNorm1d
        torch.manual_seed(1)
        bn = NaiveSyncBatchNorm1d(128, momentum=0.1).to(DEV).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        s, dy = (t.to(DEV) for t in _rows(10 + rank))
        y, prep, cfg = B._bn_rows_fwd(bn, s)
        assert cfg == (True, float(world * s.size(0)), True)
        ds, dw, db = B._bn_rows_bwd(dy, s, prep, cfg)
        ret[rank] = dict(y=y.cpu(), ds=ds.cpu(), dw=dw.cpu(), db=db.cpu(), w=bn.weight.detach().cpu(), b=bn.bias.detach().cpu(),
                         rm=bn.running_mean.cpu(), rv=bn.running_var.cpu())
    finally:
        dist.destroy_process_group()


def test_batch_norm_rows_under_sync_bn_equal_one_float64_batch():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    s = torch.cat([_rows(10 + r)[0] for r in range(world)]).double().requires_grad_(True)
    dy = torch.cat([_rows(10 + r)[1] for r in range(world)]).double()
    w, b = ret[0]['w'].double().requires_grad_(True), ret[0]['b'].double().requires_grad_(True)
    mean, var = s.mean(0), s.var(0, unbiased=False)
    y = (s - mean) * torch.rsqrt(var + 1e-5) * w + b
    (y * dy).sum().backward()
    n = s.size(0) // world
    for r in range(world):
        sl = slice(r * n, (r + 1) * n)
        assert float((ret[r]['y'].double() - y.detach()[sl]).abs().max()) < 1e-4
        assert float((ret[r]['ds'].double() - s.grad[sl]).abs().max()) < 1e-4
    # the parameter gradients stay per rank (the data-parallel all-reduce sums them afterwards): their sum is the joint gradient
    assert float((sum(ret[r]['dw'].double() for r in range(world)) - w.grad).abs().max()) < 1e-3 * float(w.grad.abs().max())
    assert float((sum(ret[r]['db'].double() for r in range(world)) - b.grad).abs().max()) < 1e-3 * float(b.grad.abs().max())
    assert torch.equal(ret[0]['rm'], ret[1]['rm']) and torch.equal(ret[0]['rv'], ret[1]['rv'])
    assert float((ret[0]['rm'].double() - 0.1 * mean.detach()).abs().max()) < 1e-5     # running += momentum * (stat - running)
