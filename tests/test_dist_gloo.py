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


def _worker4(rank, world, port, ret):
    """world 4: every rank lacks the gradients of a DIFFERENT head, a naiveSyncBN layer puts blocking
    collectives of the default communicator into the same backward pass, and both ReduceOp.AVG branches are driven."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sst_amd import parallel
        from sst_amd.norm import NaiveSyncBatchNorm1d
        from sst_amd.parallel import GradBucketReducer
        torch.manual_seed(0)
        full = torch.randn(4 * 24, 8)
        xin = full[24 * rank:24 * rank + 12 + 3 * rank]              # different numbers of rows per rank
        torch.manual_seed(3)
        trunk = torch.nn.Sequential(torch.nn.Linear(8, 8), NaiveSyncBatchNorm1d(8, eps=1e-3, momentum=0.01)).train()
        heads = torch.nn.ModuleList([torch.nn.Linear(8, 4) for _ in range(world)])
        ps = list(trunk.parameters()) + list(heads.parameters())

        def step(red):
            for p in ps:
                p.grad = None
            feat = trunk(xin)
            sum((heads[k](feat) ** 2).sum() for k in range(world) if k != rank).backward()     # rank r never uses head r
            local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in ps]
            assert heads[rank].weight.grad is None
            red.finish()
            for p, g_local in zip(ps, local):
                every = [torch.zeros_like(g_local) for _ in range(world)]
                dist.all_gather(every, g_local)
                assert torch.allclose(p.grad, sum(every) / world, atol=1e-6)
            return red.flat.clone()

        outs, orders = {}, {}
        groups = set()
        for tag, kw in (('in_finish', dict(static_graph=False)), ('overlap', dict(static_graph=True))):
            red = GradBucketReducer(ps, n_buckets=4, overlap=True, **kw)
            groups.add(id(red.group))
            natural_avg = red._avg                       # whatever this build's gloo answers to the ReduceOp.AVG probe
            assert red.overlap == (tag == 'overlap')
            sent = []
            orig = red._send
            red._send = lambda b, _o=orig, _s=sent: (_s.append((b, torch.is_grad_enabled())), _o(b))[1]
            outs[tag] = step(red)
            orders[tag] = [b for b, _ in sent]
            n_buckets = len(red.buckets)
            red.remove()
        assert len(groups) == 1, 'one bucket communicator per process, shared by the reducers'
        assert n_buckets >= 3 and orders['in_finish'] == orders['overlap'] == list(range(n_buckets))
        assert torch.allclose(outs['in_finish'], outs['overlap'], atol=1e-7)
        # both ReduceOp.AVG branches, whatever the build does by itself: a backend that REFUSES the op at the call (the probe must
        # catch it and the reducer sum + divide), and one that HAS it (RCCL following NCCL >= 2.10), emulated by sum + divide on wait
        real = dist.all_reduce

        class _Avg(object):
            def __init__(self, work, t):
                self.work, self.t = work, t

            def wait(self):
                self.work.wait()
                self.t.div_(world)

        def with_avg(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
            if op == dist.ReduceOp.AVG:
                w = _Avg(real(t, op=dist.ReduceOp.SUM, group=group, async_op=True), t)
                if async_op:
                    return w
                w.wait()
                return None
            return real(t, op=op, group=group, async_op=async_op)

        def refusing(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
            if op == dist.ReduceOp.AVG:
                raise RuntimeError('this backend does not reduce with AVG')
            return real(t, op=op, group=group, async_op=async_op)

        for tag, fake, want in (('avg', with_avg, True), ('no_avg', refusing, False)):
            parallel._AVG_SUPPORT.clear()
            dist.all_reduce = fake
            try:
                red = GradBucketReducer(ps, n_buckets=4, overlap=True)
                assert red._avg is want
                outs[tag] = step(red)
                red.remove()
            finally:
                dist.all_reduce = real
                parallel._AVG_SUPPORT.clear()
        assert isinstance(natural_avg, bool)
        assert torch.allclose(outs['no_avg'], outs['overlap'], atol=1e-7)
        assert torch.allclose(outs['avg'], outs['overlap'], atol=1e-7)
        ret[rank] = outs['overlap']
    finally:
        dist.destroy_process_group()


def test_unequal_gradient_sets_world4_with_sync_bn_and_both_avg_branches():
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker4, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1, 2, 3]
    for r in range(1, world):
        assert torch.equal(ret[0], ret[r])
