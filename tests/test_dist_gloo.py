"""N > 1 path on CPU: world_size 2 over gloo.  Covers the only collectives of the path:
naiveSyncBN's all_gather(forward)/all_reduce(backward) and the flat-bucket gradient all-reduce of bench.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sst_amd.norm import NaiveSyncBatchNorm1d
        import bench
        torch.manual_seed(0)
        full = torch.randn(64, 8)
        sizes = [40, 24]                      # ranks hold different numbers of points, as in LiDAR frames
        start = sum(sizes[:rank])
        x = full[start:start + sizes[rank]].clone().requires_grad_(True)
        bn = NaiveSyncBatchNorm1d(8, eps=1e-3, momentum=0.01)
        lin = torch.nn.Linear(8, 4)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 8))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, 8))
        model = torch.nn.Sequential(bn, lin)
        model.train()
        y = model(x)
        (y ** 2).sum().backward()
        params = [p for p in model.parameters() if p.requires_grad]
        local = torch.cat([p.grad.reshape(-1) for p in params]).clone()
        flat = bench.allreduce_grads(params, world)
        assert all(p.grad.data_ptr() >= flat.data_ptr() for p in params)  # .grad are views of the bucket
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(flat, sum(gathered) / world, atol=1e-6)
        ret[rank] = dict(y=y.detach(), gx=x.grad.detach(), flat=flat.clone(), mean=bn.running_mean.clone(),
                         gw=bn.weight.grad.clone())
        # the bucketed reducer (sst_amd/parallel.py): same averaged gradients, persistent buffer, buckets sent from hooks
        from sst_amd.parallel import GradBucketReducer
        torch.manual_seed(1)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                  torch.nn.Linear(16, 4))
        unused = torch.nn.Linear(3, 3)                   # never part of the graph: must arrive as zeros
        ps = list(net.parameters()) + list(unused.parameters())
        results = {}
        for mode, kw in (('overlap3', dict(n_buckets=3, overlap=True)), ('flat', dict(n_buckets=1, overlap=False))):
            red = GradBucketReducer(ps, **kw)
            ptr = red.flat.data_ptr()
            for it in range(2):                          # two steps: the buffer and the views persist
                for p in ps:
                    p.grad = None
                xin = full[start:start + sizes[rank]] * (it + 1)
                (net(xin) ** 2).sum().backward()
                local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in ps]
                out = red.finish()
                assert out.data_ptr() == ptr and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ps, red.views))
                for p, g_local in zip(ps, local):
                    both = [torch.zeros_like(g_local) for _ in range(world)]
                    dist.all_gather(both, g_local)
                    assert torch.allclose(p.grad, sum(both) / world, atol=1e-6)
            assert sum(e - s_ for s_, e, _ in red.buckets) == red.flat.numel() == sum(p.numel() for p in ps)
            results[mode] = red.flat.clone()
            red.remove()
        assert torch.equal(results['overlap3'], results['flat'])
        assert float(results['flat'][:12].abs().sum()) == 0.0      # the unused layer (registered last) sits first
        # the set of parameters WITHOUT a gradient differs between the ranks (a class head without points on
        # one of them).  Rank 0 skips the LAST layer's branch, rank 1 the FIRST one's: without the strict index order rank 0
        # would put its bucket 1 on the wire mid-backward while rank 1 holds it back until finish() - collectives matched by
        # order would pair different buckets.  Both orders must be 0, 1, 2 and the averages right.
        torch.manual_seed(2)
        heads = torch.nn.ModuleList([torch.nn.Linear(8, 4) for _ in range(3)])
        hp = list(heads.parameters())
        red = GradBucketReducer(hp, n_buckets=3, overlap=True)
        sent = []
        orig_send = red._send
        red._send = lambda b: (sent.append(b), orig_send(b))[1]
        for p in hp:
            p.grad = None
        xin = full[start:start + sizes[rank]]
        use = [0, 1] if rank == 0 else [1, 2]
        sum((heads[k](xin) ** 2).sum() for k in use).backward()
        local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in hp]
        red.finish()
        assert sent == [0, 1, 2], sent
        first_order = list(sent)
        for p, g_local in zip(hp, local):
            both = [torch.zeros_like(g_local) for _ in range(world)]
            dist.all_gather(both, g_local)
            assert torch.allclose(p.grad, sum(both) / world, atol=1e-6)
        # a second backward pass before finish() must not lose gradients silently
        for p in hp:
            p.grad = None
        sum((heads[k](xin) ** 2).sum() for k in range(3)).backward()
        try:
            sum((heads[k](xin) ** 2).sum() for k in range(3)).backward()
            raised = False
        except RuntimeError as e:
            raised = 'one backward pass per finish' in str(e)
        assert raised
        red._work, red._sent, red._next = [w.wait() for w in red._work] and [], [False] * 3, 0   # drain the aborted step
        red._pending = [len(idx) for _, _, idx in red.buckets]
        red.remove()
        ret[f'order{rank}'] = first_order
    finally:
        dist.destroy_process_group()


def test_sync_bn_and_flat_grad_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert 0 in ret and 1 in ret and ret['order0'] == ret['order1'] == [0, 1, 2]
    # the flat gradient bucket is identical on both ranks after the all-reduce
    assert torch.equal(ret[0]['flat'], ret[1]['flat'])
    assert torch.equal(ret[0]['gw'], ret[1]['gw'])
    # naiveSyncBN semantics (ops/norm.py:54-86): statistics = plain average over ranks of the per-rank
    # mean / mean-of-squares (NOT weighted by point count)
    torch.manual_seed(0)
    full = torch.randn(64, 8)
    parts = [full[:40], full[40:]]
    mean = sum(p.mean(0) for p in parts) / 2
    meansqr = sum((p * p).mean(0) for p in parts) / 2
    var = meansqr - mean * mean
    w, b = torch.linspace(0.5, 1.5, 8), torch.linspace(-0.2, 0.2, 8)
    lin = None
    for r, p in enumerate(parts):
        ref = (p - mean) * torch.rsqrt(var + 1e-3) * w + b
        torch.manual_seed(0)
        _ = torch.randn(64, 8)
        if lin is None:
            # same construction order as the worker: bn first, then the Linear draws its init from the RNG
            from sst_amd.norm import NaiveSyncBatchNorm1d
            NaiveSyncBatchNorm1d(8)
            lin = torch.nn.Linear(8, 4)
        assert torch.allclose(ret[r]['y'], lin(ref), atol=1e-5)
    assert torch.allclose(ret[0]['mean'], 0.01 * mean, atol=1e-6)
    assert torch.isfinite(ret[0]['gx']).all() and ret[0]['gx'].shape == (40, 8)
