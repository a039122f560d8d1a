#!/usr/bin/env python
"""Developer microbenchmark (GPU box): times individual kernels of the path on the bench workload with HIP
events (interleaved rounds, median) and prints achieved GB/s against their algorithmic bytes.
Usage: python tools/microbench.py [sra|ln|unique|all]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sst_amd  # noqa: E402
from sst_amd import kernels as K  # noqa: E402

DEV = torch.device('cuda:0')


def timeit(fn, iters=30, warmup=5, reps=1):
    """median / min ms per call; reps > 1 queues that many calls between the two events (kernels shorter than the
    host's launch time are otherwise timed as launch gaps)"""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts)), float(np.min(ts))


def frame_plan(train=True):
    pts = bench.make_cloud(116000, 0, DEV)
    vox = sst_amd.Voxelization(bench.VOXEL_SIZE, bench.PC_RANGE, -1, (-1, -1))
    _, coors = vox.voxelize_batch([pts])
    sp = sst_amd.build_scatter_plan(coors, grid_zyx=[1, 468, 468])
    layer = sst_amd.SSTInputLayerV2((bench.DROP_TRAIN, bench.DROP_TEST), (12, 12, 1), (468, 468, 1),
                                    shuffle_voxels=False, mute=True, reference_outputs=False, debug=False)
    layer.train(train)
    feats = torch.randn(sp.num_voxels, 128, device=DEV)
    return pts, coors, layer(feats, sp.voxel_coors, 1)


def bench_sra():
    _, _, info = frame_plan()
    m = info['voxel_feats'].size(0)
    for s in range(2):
        plan = info[f'sra_plan_shift{s}']
        off = plan.winoff[:plan.n_windows + 1].cpu().numpy()
        sizes = np.diff(off)
        print(f'shift{s}: {plan.n_windows} windows, tokens {m}, size min/mean/max {sizes.min()}/{sizes.mean():.1f}/'
              f'{sizes.max()}, tiles hist {np.bincount((sizes + 15) // 16).tolist()}')
        qk = torch.randn(m, 256, device=DEV)
        v = torch.randn(m, 128, device=DEV)
        do = torch.randn(m, 128, device=DEV)
        for impl in (0, 2):
            med, mn = timeit(lambda: K._sra_fwd(qk[:, :128], qk[:, 128:], v, plan, 8, 0.25, impl))
            gbs = bench.SRA_BYTES_PER_TOKEN * m / (med * 1e-3) / 1e9
            print(f'  fwd impl={impl}: median {med * 1e3:.1f} us (min {mn * 1e3:.1f}) -> {gbs:.0f} GB/s '
                  f'({gbs / 80:.1f} % of 8 TB/s)')
        o, lse = K._sra_fwd(qk[:, :128], qk[:, 128:], v, plan, 8, 0.25, 0)
        dqk = torch.empty_like(qk)
        dv = torch.empty_like(v)
        plan100 = K.WindowPlan(plan.tok, plan.winoff, plan.n_windows, plan.n_tokens, 100)  # announce the cap only
        print(f'  plan.max_tokens = {plan.max_tokens} (largest window); second plan announces 100')
        for impl, pl, tag in ((0, plan, 'one-pass, class by largest window'), (0, plan100, 'one-pass, 7-tile class'),
                              (3, plan, 'two launches'), (0, plan, 'one-pass, class by largest window'),
                              (0, plan100, 'one-pass, 7-tile class'), (3, plan, 'two launches')):
            med, mn = timeit(lambda: K._sra_bwd(qk[:, :128], qk[:, 128:], v, o, lse, do, pl, 8, 0.25, impl,
                                                dqk[:, :128], dqk[:, 128:], dv))
            gbs = bench.SRA_BWD_BYTES_PER_TOKEN * m / (med * 1e-3) / 1e9
            print(f'  bwd impl={impl} ({tag}): median {med * 1e3:.1f} us (min {mn * 1e3:.1f}) -> {gbs:.0f} GB/s '
                  f'({gbs / 80:.1f} % of 8 TB/s)')


def bench_ln():
    from sst_amd.dense import add_layer_norm, colsum, weight_grad_splitk
    m, c = 90107, 128
    x = torch.randn(m, c, device=DEV, requires_grad=True)
    r = torch.randn(m, c, device=DEV, requires_grad=True)
    norm = torch.nn.LayerNorm(c).to(DEV)
    g = torch.randn(m, c, device=DEV)
    med, _ = timeit(lambda: add_layer_norm(x, r, norm))
    print(f'add+LN fwd: {med * 1e3:.1f} us -> {(3 * m * c * 4 + m * 8) / med / 1e6:.0f} GB/s')
    y = add_layer_norm(x, r, norm)
    med, _ = timeit(lambda: torch.autograd.grad(y, (x, r), g, retain_graph=True))
    print(f'add+LN bwd: {med * 1e3:.1f} us -> {(3 * m * c * 4) / med / 1e6:.0f} GB/s')
    med, _ = timeit(lambda: torch.nn.functional.layer_norm(x + r, (c,), norm.weight, norm.bias))
    print(f'torch add + layer_norm fwd: {med * 1e3:.1f} us')
    dy = torch.randn(m, 384, device=DEV)
    xx = torch.randn(m, 128, device=DEV)
    med, _ = timeit(lambda: dy.t() @ xx)
    print(f'dW plain  [384x{m}]x[{m}x128]: {med * 1e3:.1f} us')
    for chunk in (512, 1024, 2048, 4096):
        med, _ = timeit(lambda: weight_grad_splitk(dy, xx, chunk))
        print(f'dW split-K chunk {chunk}: {med * 1e3:.1f} us')
    med, _ = timeit(lambda: colsum(dy))
    print(f'colsum [{m},384]: {med * 1e3:.1f} us; torch sum(0): {timeit(lambda: dy.sum(0))[0] * 1e3:.1f} us')
    w = torch.randn(384, 128, device=DEV)
    med, _ = timeit(lambda: xx @ w.t())
    print(f'fwd GEMM [{m}x128]x[128x384]: {med * 1e3:.1f} us -> {2 * m * 128 * 384 / med / 1e9:.1f} TFLOP/s')


def bench_unique():
    pts, coors, info = frame_plan()
    med, _ = timeit(lambda: K.dynamic_voxelize(pts, bench.VOXEL_SIZE, bench.PC_RANGE), iters=50)
    print(f'dynamic_voxelize 116k: {med * 1e3:.1f} us -> {24 * 116000 / med / 1e6:.1f} GB/s')
    med, _ = timeit(lambda: K.unique_rows(coors, [0, -1, -1, -1], [1, 2, 469, 469], invalid_if_negative=2))
    print(f'unique_rows (incl. host readback): {med * 1e3:.1f} us')
    vc = info['voxel_coors'].contiguous()
    w0, c0, w1, c1 = K.window_coors(vc, [468, 468, 1], [12, 12, 1])
    levels = [(30, 0, 30), (60, 30, 60), (100, 60, 100000)]
    med, _ = timeit(lambda: K.region_batching(w0, w1, 13, levels))
    print(f'region_batching: {med * 1e3:.1f} us')
    layer = sst_amd.SSTInputLayerV2((bench.DROP_TRAIN, bench.DROP_TEST), (12, 12, 1), (468, 468, 1),
                                    shuffle_voxels=True, mute=True, reference_outputs=False, debug=False)
    layer.train()
    feats = torch.randn(vc.size(0), 128, device=DEV)
    med, _ = timeit(lambda: layer(feats, vc, 1))
    print(f'SSTInputLayerV2 forward (shuffle, no reference dicts): {med * 1e3:.1f} us')


def bench_gemm():
    """Alternatives for the tall weight-gradient GEMM dW[out,in] = dY[M,out]^T X[M,in] and for column sums."""
    from sst_amd.dense import colsum
    m = 90107
    for out, inn in ((384, 128), (256, 128), (128, 128), (256, 128), (128, 256)):
        dy = torch.randn(m, out, device=DEV)
        x = torch.randn(m, inn, device=DEV)
        ref = dy.double().t() @ x.double()
        flops = 2.0 * m * out * inn

        def report(name, fn):
            med, _ = timeit(fn, iters=15, warmup=3)
            err = float((fn().double() - ref).abs().max() / ref.abs().max())
            print(f'  dW[{out}x{inn}] {name:34s} {med * 1e3:7.1f} us  {flops / med / 1e9:6.1f} TF/s  rel.err {err:.1e}')

        report('dy.t() @ x', lambda: dy.t() @ x)
        report('(x.t() @ dy).t()', lambda: (x.t() @ dy).t())
        for chunk in (1024, 4096, 8192):
            s_ = m // chunk
            body = s_ * chunk

            def splitk(chunk=chunk, s_=s_, body=body):
                dw = torch.bmm(dy[:body].view(s_, chunk, out).transpose(1, 2), x[:body].view(s_, chunk, inn)).sum(0)
                return dw + dy[body:].t() @ x[body:]

            def splitk_t(chunk=chunk, s_=s_, body=body):
                dw = torch.bmm(x[:body].view(s_, chunk, inn).transpose(1, 2), dy[:body].view(s_, chunk, out)).sum(0)
                return (dw + x[body:].t() @ dy[body:]).t()
            report(f'bmm split-K chunk {chunk}', splitk)
            report(f'bmm split-K (x^T dy) chunk {chunk}', splitk_t)
        from sst_amd.dense import weight_bias_grad
        report('wgrad kernel (+ bias grad)', lambda: weight_bias_grad(dy, x, True)[0])
    dy = torch.randn(m, 384, device=DEV)
    ones = torch.ones(m, device=DEV)
    for name, fn in (('colsum kernel', lambda: colsum(dy)), ('dy.sum(0)', lambda: dy.sum(0)),
                     ('torch.mv(dy.t(), ones)', lambda: torch.mv(dy.t(), ones)),
                     ('ones[None] @ dy', lambda: ones[None] @ dy)):
        med, _ = timeit(fn, iters=15, warmup=3)
        print(f'  colsum[{m}x384] {name:28s} {med * 1e3:7.1f} us  {m * 384 * 4 / med / 1e6:7.0f} GB/s')
    # forward-shaped GEMMs
    x = torch.randn(m, 128, device=DEV)
    for out in (128, 256, 384):
        w = torch.randn(out, 128, device=DEV)
        b = torch.randn(out, device=DEV)
        med, _ = timeit(lambda: torch.addmm(b, x, w.t()), iters=15, warmup=3)
        print(f'  fwd addmm [{m}x128]x[128x{out}]: {med * 1e3:7.1f} us  {2.0 * m * 128 * out / med / 1e9:6.1f} TF/s')


def bench_sir():
    """FSD's SIR point-group backbone (configs/fsd: 3 SIRLayer blocks, LN + GELU, max pooling) on a synthetic
    foreground set: 30 000 points in ~3 000 clusters.  Times forward + backward and the
    segmented-max kernel against its algorithmic bytes."""
    torch.manual_seed(0)
    n, n_clusters = 30000, 3000
    sir = sst_amd.build_backbone(dict(type='SIR', num_blocks=3, in_channels=[84, 133, 133],
                                      feat_channels=[[128, 128]] * 3, rel_mlp_hidden_dims=[[16, 32]] * 3,
                                      norm_cfg=dict(type='LN', eps=1e-3), mode='max', xyz_normalizer=[20, 20, 4],
                                      act='gelu', unique_once=True)).to(DEV).train()
    cluster = torch.randint(0, n_clusters, (n,), device=DEV)
    coors = torch.stack([torch.zeros_like(cluster), torch.zeros_like(cluster), cluster], 1)  # int64 (cls, batch, id)
    points = torch.randn(n, 5, device=DEV) * 3
    feats = torch.randn(n, 79, device=DEV).requires_grad_(True)  # 79 + xyz + 2 = in_channels 84 as in the golden case
    f_cluster = torch.randn(n, 3, device=DEV)

    def fwd_bwd():
        feats.grad = None
        p, c, _ = sir(points, feats, coors, f_cluster)
        (p.sum() + c.sum()).backward()

    med, _ = timeit(fwd_bwd, iters=20, warmup=3)
    print(f'SIR x3 (30k points, {n_clusters} clusters) fwd+bwd: {med * 1e3:.1f} us')
    with torch.no_grad():
        med, _ = timeit(lambda: sir(points, feats, coors, f_cluster), iters=20, warmup=3)
    print(f'SIR x3 forward: {med * 1e3:.1f} us')
    # the pooling kernel alone, at FSD size and at the VFE size of the SST bench
    for npts, c, groups in ((30000, 128, 3000), (116000, 128, 90000), (300000, 128, 60000)):
        idx = torch.randint(0, groups, (npts,), device=DEV)
        plan = K.unique_rows(idx[:, None].int().contiguous(), [0], [groups])
        x = torch.randn(npts, c, device=DEV)
        med, _ = timeit(lambda: K.segment_reduce(x, plan, 'max'), iters=30, warmup=3)
        byt = npts * (4 * c + 4) + plan.m * 4 * c
        print(f'segment max [{npts} x {c}] -> {plan.m} groups: {med * 1e3:.1f} us -> {byt / med / 1e6:.0f} GB/s '
              f'({100 * byt / med / 1e6 / 8000:.1f} % of 8 TB/s)')


def bench_cluster():
    """FSD cluster assignment: the union-find kernel against the reference's data flow on the same box (dense
    N x N distance matrix on the GPU -> .cpu() -> scipy connected_components -> .to(device)), labels compared."""
    import time
    from scipy.sparse.csgraph import connected_components
    for n, dist, spread in ((2000, 0.6, 60.0), (8000, 0.6, 120.0), (20000, 0.3, 150.0)):
        g = torch.Generator().manual_seed(n)
        pts = (torch.rand(n, 3, generator=g) * spread).to(DEV)
        batch = torch.sort(torch.randint(0, 2, (n,), generator=g))[0].int().to(DEV)
        med, _ = timeit(lambda: sst_amd.find_connected_componets(pts, batch, dist), iters=20, warmup=3)
        got = sst_amd.find_connected_componets(pts, batch, dist)

        def reference_flow():
            out = torch.zeros_like(batch) - 1
            base = 0
            for i in range(int(batch.max().item()) + 1):
                m = batch == i
                p = pts[m]
                d = p[:, None, :2] - p[None, :, :2]
                adj = ((d ** 2).sum(2) ** 0.5 < dist).cpu().numpy()
                c = torch.from_numpy(connected_components(adj, directed=False)[1]).to(DEV).int() + base
                base = int(c.max().item()) + 1
                out[m] = c
            return out
        reference_flow()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = reference_flow()
        torch.cuda.synchronize()
        t_ref = time.perf_counter() - t0
        same = bool(torch.equal(ref, got))
        print(f'connected components n={n} dist={dist}: kernel {med * 1e3:.0f} us ({int(got.max()) + 1} components), '
              f'reference data flow {t_ref * 1e3:.1f} ms, labels identical: {same}')


def bench_voxelize():
    """dynamic_voxelize at sizes where the HBM roof matters (a frame is 2.8 MB: launch-bound): algorithmic 24 B/point
    (SURVEY.md §8d); the bytes really moved are 4*C read + 16 written per point for [N, C] points and (b, z, y, x) rows."""
    vox = sst_amd.Voxelization(bench.VOXEL_SIZE, bench.PC_RANGE, -1, (-1, -1))
    for n, c in ((116000, 5), (1160000, 5), (16000000, 5), (16000000, 4), (16000000, 3)):
        pts = torch.rand(n, c, device=DEV) * 150 - 75
        med, mn = timeit(lambda: vox.voxelize_batch([pts]), iters=20, warmup=3)
        print(f'dynamic_voxelize {n} points x {c} floats: {med * 1e3:.1f} us (min {mn * 1e3:.1f}) -> algorithmic '
              f'{24 * n / med / 1e6:.0f} GB/s ({100 * 24 * n / med / 1e6 / 8000:.1f} % of 8 TB/s), moved '
              f'{(4 * c + 16) * n / med / 1e6:.0f} GB/s')


def bench_spconv():
    """sparse convolution at FSD scale: rulebook construction, the gathered-GEMM forward / data gradient, the
    pair-list weight gradient, against the reference's formulation in torch on the same box (per kernel offset:
    gather, mm, index_add -- spconv_ops.h:305-350).  Useful flops = 2 * pairs * Cin * Cout."""
    import numpy as np
    from sst_amd import spconv
    rng = np.random.default_rng(8)
    batch, shape = 2, [41, 800, 800]
    for n_target, cin, cout in ((150000, 64, 64), (150000, 128, 128), (60000, 64, 128)):
        vol = int(np.prod([shape[0], shape[1] // 2, shape[2] // 2]))
        lin = rng.choice(batch * vol, n_target // 4, replace=False)
        b, r = lin // vol, lin % vol
        hs = [shape[0], shape[1] // 2, shape[2] // 2]
        base = np.stack([b, r // (hs[1] * hs[2]), (r // hs[2]) % hs[1], r % hs[2]], 1)
        ind = np.unique(np.concatenate([base * [1, 1, 2, 2] + [0, 0, dy, dx] for dy in (0, 1) for dx in (0, 1)]), axis=0)
        ind = torch.from_numpy(ind.astype(np.int32)).to(DEV)
        n = ind.size(0)
        med_rb, _ = timeit(lambda: spconv.get_indice_pairs(ind, batch, shape, 3, subm=True), iters=10, warmup=2)
        outids, pairs, num = spconv.get_indice_pairs(ind, batch, shape, 3, subm=True)
        med_rb2, _ = timeit(lambda: spconv.get_indice_pairs(ind, batch, shape, 3, 2, 1), iters=10, warmup=2)
        npairs = int(num.sum())
        x = torch.randn(n, cin, device=DEV)
        w = torch.randn(3, 3, 3, cin, cout, device=DEV) * 0.05
        gy = torch.randn(n, cout, device=DEV)
        med_f, _ = timeit(lambda: spconv.indice_conv(x, w, pairs, num, n, False, True), iters=20, warmup=3)
        med_b, _ = timeit(lambda: spconv.indice_conv_backward(x, w, gy, pairs, num, False, True), iters=20, warmup=3)
        rb = pairs._sst_rulebook
        w3c = w.view(27, cin, cout)
        med_d, _ = timeit(lambda: spconv._gather_gemm(gy, rb.in2out, n, w3c, True, cin), iters=20, warmup=3)
        shuf = ind[torch.randperm(n, device=DEV)]
        _, pairs_s, num_s = spconv.get_indice_pairs(shuf, batch, shape, 3, subm=True)
        med_fs, _ = timeit(lambda: spconv.indice_conv(x, w, pairs_s, num_s, n, False, True), iters=20, warmup=3)
        w3 = w.view(27, cin, cout)
        counts = num.tolist()

        def reference_flow():
            out = torch.zeros(n, cout, device=DEV)
            for k in range(27):
                c = counts[k]
                if c:
                    out.index_add_(0, pairs[k, 1, :c].long(), x[pairs[k, 0, :c].long()] @ w3[k])
            return out
        med_r, _ = timeit(reference_flow, iters=10, warmup=2)
        fl = 2.0 * npairs * cin * cout
        print(f'spconv SubM3 {n} voxels {cin}->{cout}, {npairs} pairs ({npairs / n:.1f} per voxel): rulebook {med_rb * 1e3:.0f} us '
              f'(stride-2 conv rulebook {med_rb2 * 1e3:.0f} us); forward {med_f * 1e3:.0f} us = {fl / med_f / 1e9:.1f} useful TFLOP/s '
              f'({27 * 2.0 * n * cin * cout / med_f / 1e9:.1f} if nothing were skipped; voxels in (b, z, y, x) order; {med_fs * 1e3:.0f} us in random order); '
              f'dgrad {med_d * 1e3:.0f} us, dgrad + wgrad {med_b * 1e3:.0f} us; '
              f'per-offset gather/mm/index_add forward in torch {med_r * 1e3:.0f} us')


def bench_unet():
    """FSD's SimpleSparseUNet (configs/fsd/fsd_waymoD1_1x.py:39-51) forward + backward on a synthetic 2-sample
    batch of ~120 k voxels."""
    import numpy as np
    net = sst_amd.BACKBONES.build(dict(
        type='SimpleSparseUNet', in_channels=64, sparse_shape=[32, 640, 640], order=('conv', 'norm', 'act'),
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), base_channels=64, output_channels=128,
        encoder_channels=((64, ), (64, 64, 64), (64, 64, 64), (128, 128, 128), (256, 256, 256)),
        encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1), (1, 1, 1)),
        decoder_channels=((256, 256, 128), (128, 128, 64), (64, 64, 64), (64, 64, 64), (64, 64, 64)),
        decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1)))).to(DEV)
    rng = np.random.default_rng(2)
    hs = [16, 320, 320]
    vol = int(np.prod(hs))
    lin = rng.choice(2 * vol, 30000, replace=False)
    b, r = lin // vol, lin % vol
    base = np.stack([b, r // (hs[1] * hs[2]), (r // hs[2]) % hs[1], r % hs[2]], 1)
    ind = np.unique(np.concatenate([base * [1, 2, 2, 2] + [0, 0, dy, dx] for dy in (0, 1) for dx in (0, 1)]), axis=0)
    ind = torch.from_numpy(ind.astype(np.int32)).to(DEV)
    x = torch.randn(ind.size(0), 64, device=DEV, requires_grad=True)

    def fwd_bwd():
        x.grad = None
        for p in net.parameters():
            p.grad = None
        out = net({'voxel_feats': x, 'voxel_coors': ind})[0]
        out['voxel_feats'].sum().backward()

    med, _ = timeit(fwd_bwd, iters=10, warmup=3)
    with torch.no_grad():
        med_f, _ = timeit(lambda: net({'voxel_feats': x, 'voxel_coors': ind}), iters=10, warmup=3)
    print(f'SimpleSparseUNet (FSD config) on {ind.size(0)} voxels: forward {med_f:.2f} ms, forward + backward {med:.2f} ms')


def bench_pointpool():
    """dynamic point pool at FSD second-stage sizes: whole op (3 passes + scans + the count read-back) and the pair
    tests per second it amounts to (the reference's kernel is the same R x P brute force, one thread per pair)."""
    import numpy as np
    for n_rois, n_pts, max_all in ((300, 50000, 50000), (1000, 100000, 100000), (2000, 200000, 100000)):
        rng = np.random.default_rng(n_rois)
        rois = np.concatenate([rng.uniform(-70, 70, (n_rois, 2)), rng.uniform(-2, 1, (n_rois, 1)),
                               rng.uniform(0.5, 5, (n_rois, 3)), rng.uniform(-4, 4, (n_rois, 1))], 1).astype(np.float32)
        k = rng.integers(0, n_rois, n_pts)
        pts = (rois[k, :3] + rng.normal(0, 1.0, (n_pts, 3)) + np.array([0, 0, 1.0])).astype(np.float32)
        r, p = torch.from_numpy(rois).to(DEV), torch.from_numpy(pts).to(DEV)
        med, _ = timeit(lambda: sst_amd.dynamic_point_pool(r, p, [0.5, 0.5, 0.5], 256, max_all), iters=20, warmup=3)
        n_out = len(sst_amd.dynamic_point_pool(r, p, [0.5, 0.5, 0.5], 256, max_all)[0])
        print(f'dynamic point pool {n_rois} RoIs x {n_pts} points -> {n_out} pairs: {med * 1e3:.0f} us '
              f'({2 * n_rois * n_pts / med / 1e6:.1f} G pair tests/s over the two passes)')


def bench_dense_bf16(m=90107):
    """bf16 tall linears and the grouped weight gradient (csrc/dense_bf16.hip) against their algorithmic bytes, with
    the library product (torch addmm / matmul on the same bf16 tensors) beside each"""
    from sst_amd import bf16
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    mk = lambda c: torch.randn(m, c, generator=g).to(BF).to(DEV)
    for k, n in ((128, 128), (128, 256), (256, 128)):
        x, aux = mk(k), mk(n)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(BF).to(DEV)
        b = torch.randn(n, generator=g).to(DEV)
        bb = b.to(BF)
        nbytes = m * (k + n) * 2
        for name, fn, extra in (('bias', lambda: bf16.tall_linear(x, w, b), 0),
                                ('gelu+pre', lambda: bf16.tall_linear(x, w, b, bf16.EPI_GELU, want_pre=True), m * n * 2),
                                ('*gelu_grad', lambda: bf16.tall_linear(x, w, None, bf16.EPI_MUL_GELU_GRAD, aux_in=aux), m * n * 2),
                                ('+add', lambda: bf16.tall_linear(x, w, None, bf16.EPI_ADD, aux_in=aux), m * n * 2),
                                ('library addmm', lambda: torch.addmm(bb, x, w.t()), 0)):
            med, mn = timeit(fn, iters=10, reps=20)
            print(f'tall_linear_bf16 K={k} N={n} {name:14s}: {med * 1e3:7.1f} us (min {mn * 1e3:6.1f})  '
                  f'{(nbytes + extra) / med / 1e6:7.1f} GB/s of {(nbytes + extra) / 1e6:.1f} MB')
    dqkv, xp, x, ds1, o, dpre, y1, h, ds2 = mk(384), mk(128), mk(128), mk(128), mk(128), mk(256), mk(128), mk(256), mk(128)
    f32 = dict(dtype=torch.float32, device=DEV)
    dw_in, db_in = torch.empty((384, 128), **f32), torch.empty(384, **f32)
    dwo, dbo = torch.empty((128, 128), **f32), torch.empty(128, **f32)
    dw1, db1 = torch.empty((256, 128), **f32), torch.empty(256, **f32)
    dw2, db2 = torch.empty((128, 256), **f32), torch.empty(128, **f32)
    probs = [(dqkv[:, :256], xp, dw_in[:256], db_in[:256], 1, 0), (dqkv[:, 256:], x, dw_in[256:], db_in[256:], 1, 0),
             (ds1, o, dwo, dbo, 1, 0), (dpre, y1, dw1, db1, 1, 0), (h, ds2, dw2, db2, 2, 1)]
    nbytes = m * (384 + 128 + 128 + 128 + 128 + 256 + 128 + 256 + 128) * 2
    med, mn = timeit(lambda: bf16.wgrad_group(probs))
    print(f'wgrad_group_bf16 (5 products of one layer + bias sums): {med * 1e3:7.1f} us (min {mn * 1e3:6.1f})  '
          f'{nbytes / med / 1e6:7.1f} GB/s of {nbytes / 1e6:.1f} MB operand bytes')
    med, mn = timeit(lambda: [bf16.weight_grad(a if not tr else b, b if not tr else a) for a, b, _, _, _, tr in probs])
    print(f'  library form (batched split-K bmm + fp32 sum), same 5 products: {med * 1e3:7.1f} us')


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('dense_bf16',):
        bench_dense_bf16()
    if what in ('sra', 'all'):
        bench_sra()
    if what in ('ln', 'all'):
        bench_ln()
    if what in ('unique', 'all'):
        bench_unique()
    if what in ('gemm',):
        bench_gemm()
    if what in ('sir',):
        bench_sir()
    if what in ('cluster',):
        bench_cluster()
    if what in ('voxelize',):
        bench_voxelize()
    if what in ('unet',):
        bench_unet()
    if what in ('spconv',):
        bench_spconv()
    if what in ('pointpool',):
        bench_pointpool()
