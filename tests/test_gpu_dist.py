"""GPU: the N > 1 path of bench.py on ONE device — two ranks (gloo, both on cuda:0) run bench.Pipeline on one frame each,
naiveSyncBN exchanges its statistics (ops/norm.py:9-24), the gradients go through bench.allreduce_grads; the result
must equal the single-rank run on the concatenated two-frame batch (frames shard over ranks, the only
exchanges are the BN statistics and the gradient all-reduce).  Also: `python bench.py --gpus 2` launches its own ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_POINTS = 20000      # the same number of points on both ranks: naiveSyncBN averages per-rank means unweighted
BLOCKS = 1
GRAD_TOL = 2e-3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev):
    import bench
    torch.manual_seed(0)
    model = bench.Pipeline(BLOCKS).to(dev).train()
    model.middle_encoder.shuffle_voxels = False
    return bench, model


def _loss(out):
    """A row-order independent loss whose gradient does not vanish: the plain sum (and the sum of squares) of
    LayerNorm outputs is a constant, its upstream gradient is rounding noise."""
    return torch.tanh(2 * out[:, :64]).sum() + (out[:, 64:] ** 3).sum()


def _bn_stats(model):
    return [t.detach().cpu().clone() for layer in model.voxel_encoder.vfe_layers
            for t in (layer.norm.running_mean, layer.norm.running_var)]


def _worker(rank, world, port, ret, backend='gloo', one_device=True):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.path.insert(0, ROOT)
    dev = 'cuda:0' if one_device else f'cuda:{rank}'
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from sst_amd.parallel import GradBucketReducer
        bench, model = _build(dev)
        params = [p for p in model.parameters() if p.requires_grad]
        reducer = GradBucketReducer(params, n_buckets=2)      # what bench.py --gpus N uses: buckets sent from hooks
        frame = bench.make_cloud(N_POINTS, 100 + rank, dev)
        out = model([frame])
        _loss(out).backward()
        reducer.finish()
        ret[rank] = dict(out_sum=float(out.detach().abs().double().sum()), n=int(out.size(0)),
                         grads=[p.grad.detach().cpu().clone() for p in params], bn=_bn_stats(model))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_concatenated_batch():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == 2
    # the averaged gradients are bit-identical on both ranks
    for a, b in zip(ret[0]['grads'], ret[1]['grads']):
        assert torch.equal(a, b)
    # single rank, both frames in one batch
    bench, model = _build('cuda:0')
    params = [p for p in model.parameters() if p.requires_grad]
    frames = [bench.make_cloud(N_POINTS, 100 + r, 'cuda:0') for r in range(world)]
    out = model(frames)
    _loss(out).backward()
    assert out.size(0) == ret[0]['n'] + ret[1]['n']
    tot = float(out.detach().abs().double().sum())       # (LayerNorm outputs: the plain sum is ~0)
    assert abs(tot - (ret[0]['out_sum'] + ret[1]['out_sum'])) <= 1e-5 * tot
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    errs = {}
    for n, p, g2 in zip(names, params, ret[0]['grads']):
        ref = p.grad.cpu()
        scale = max(1e-3, float(ref.abs().max()))
        errs[n] = float((world * g2 - ref).abs().max()) / scale     # all-reduce averages, the joint batch sums
    log_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(log_dir):
        with open(os.path.join(log_dir, 'dist_grad_errs.json'), 'w') as f:
            json.dump(errs, f, indent=1)
    worst = max(errs, key=errs.get)
    assert errs[worst] < GRAD_TOL, f'{worst}: relative gradient error {errs[worst]}'
    for a, b in zip(_bn_stats(model), ret[0]['bn']):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL over xGMI)')
def test_two_ranks_over_rccl_equal_the_concatenated_batch():
    """the same check with one rank per GPU over backend 'nccl' (= RCCL): runs wherever `pytest -m gpu` sees two devices
    (the driver's 8-GPU node), so RCCL initialisation, the async bucket all-reduces (ReduceOp.AVG) and naiveSyncBN's
    messages are exercised before bench.py --gpus N is"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, 'nccl', False), nprocs=world, join=True)
    assert len(ret) == 2
    for a, b in zip(ret[0]['grads'], ret[1]['grads']):
        assert torch.equal(a, b)
    bench, model = _build('cuda:0')
    params = [p for p in model.parameters() if p.requires_grad]
    frames = [bench.make_cloud(N_POINTS, 100 + r, 'cuda:0') for r in range(world)]
    out = model(frames)
    _loss(out).backward()
    assert out.size(0) == ret[0]['n'] + ret[1]['n']
    for n, p, g2 in zip([n for n, p in model.named_parameters() if p.requires_grad], params, ret[0]['grads']):
        ref = p.grad.cpu()
        assert float((world * g2 - ref).abs().max()) / max(1e-3, float(ref.abs().max())) < GRAD_TOL, n


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the script starts its ranks itself
    (the reference: tools/dist_train.sh:7-9) and rank 0 prints the JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--share-device', '--backend',
                        'gloo', '--steps', '2', '--warmup', '1', '--points', '20000', '--blocks', '1',
                        '--no-gemm-tuning', '--no-forward-only-leg'], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['parallelism'] == 'dp2' and res['value'] > 0
    assert res['config']['grad_sync'].startswith('one flat')


def test_bench_preflight_eight_ranks_on_one_device():
    """The launch plumbing of the driver's N = 8 run without eight GPUs: eight ranks, all on cuda:0,
    gloo - rendezvous, per-rank frames, SyncBN exchange, bucketed gradient all-reduce, max-over-ranks timing, ONE JSON line
    from rank 0 with its `ranks` block."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--share-device', '--backend',
                        'gloo', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--points', '20000', '--blocks', '2'],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 8 and res['config']['parallelism'] == 'dp8' and res['value'] > 0 and res['scaling'] == 'weak'
    ranks = res['ranks']
    assert ranks['world_size'] == 8 and ranks['backend'] == 'gloo' and len(ranks['per_rank']['ms_per_step']) == 8
    assert res['config']['grad_sync'].startswith('one flat') and res['cpu_baseline'] is None
    assert 'workloads' not in res           # the single-GPU legs stay out of a multi-rank line
