#!/usr/bin/env python
"""Per-layer table of the sparse convolutions of a bench workload (GPU box): every SparseConvolution of one forward
pass of `bench.py --workload fsd|fsdv2` is captured with its rulebook, then forward, data gradient and weight gradient
are timed one by one (HIP events, median) with the output-stationary kernel (csrc/spconv_os.hip) and with the
first-generation kernels (csrc/spconv.hip).  Useful flops = 2 x pairs x Cin x Cout.
Usage: python tools/conv_layers.py [fsd|fsdv2] [points]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as BW  # noqa: E402
from sst_amd import spconv as SP  # noqa: E402

DEV = torch.device('cuda:0')
PEAK = BW.FP32_MFMA_PEAK_TFLOPS


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'fsd'
    spec = BW.WORKLOADS[what]
    n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else spec['points']
    torch.manual_seed(0)
    model = spec['cls']().to(DEV).train()
    clouds = [model.make_cloud(n_pts, 0, DEV)]
    layers = []

    def post(mod, inp, out):
        if mod.conv1x1:
            return
        datas = out.indice_dict.get(mod.indice_key) if mod.indice_key is not None else None
        if datas is None:
            return
        rb = getattr(datas[2], '_sst_rulebook', None)
        if rb is None:
            return
        layers.append(dict(mod=mod, x=inp[0].features.detach(), rb=rb, pairs=datas[2], inverse=mod.inverse,
                           kind='inv' if mod.inverse else ('subm' if mod.subm else 'conv')))

    names = {}
    for name, m in model.named_modules():
        if isinstance(m, SP.SparseConvolution):
            m.register_forward_hook(post)
            names[m] = name
    with torch.no_grad():
        model(clouds)
    torch.cuda.synchronize()
    tot = dict(fl=0.0)
    cols = ('f_os', 'f_old', 'd_os', 'd_old', 'wg', 'wg_old')
    for c in cols:
        tot[c] = 0.0
    print(f'{what}: {len(layers)} sparse convolutions, {n_pts} points')
    print(f'{"layer":44s} {"kind":4s} {"n_in":>7s} {"n_out":>7s} {"cin":>4s} {"cout":>4s} {"pairs":>8s} {"dens":>5s} | '
          f'{"fwd os":>8s} {"TF":>5s} {"fwd old":>8s} | {"dgrad os":>8s} {"dgrad old":>9s} | {"wgrad":>7s} {"TF":>5s} {"wg old":>7s}')
    for L in layers:
        mod, x, rb = L['mod'], L['x'], L['rb']
        w3 = mod.weight.detach().reshape(-1, mod.in_channels, mod.out_channels)
        cin, cout = mod.in_channels, mod.out_channels
        if L['inverse']:
            fmap, frows, dmap, drows, x_side = rb.in2out, rb.n, rb.out2in, rb.m, 1
        else:
            fmap, frows, dmap, drows, x_side = rb.out2in, rb.m, rb.in2out, rb.n, 0
        gy = torch.randn(frows, cout, device=DEV)
        res = {}
        for tag, kern in (('os', 'os'), ('old', 'legacy')):
            os.environ['SST_SPCONV_KERNEL'] = kern
            res['f_' + tag] = timeit(lambda: SP._gather_gemm(x, fmap, frows, w3, False, cout, rb))
            res['d_' + tag] = timeit(lambda: SP._gather_gemm(gy, dmap, drows, w3, True, cin, rb))
            res['wg' + ('' if tag == 'os' else '_old')] = timeit(lambda: SP._wgrad(x, gy, rb, L['pairs'], x_side, mod.weight.shape))
        os.environ['SST_SPCONV_KERNEL'] = 'os'
        fl = 2.0 * rb.total_pairs * cin * cout
        tot['fl'] += fl
        for c in cols:
            tot[c] += res[c]
        print(f'{names[mod][-44:]:44s} {L["kind"]:4s} {x.size(0):7d} {frows:7d} {cin:4d} {cout:4d} {rb.total_pairs:8d} '
              f'{rb.density:5.2f} | {res["f_os"] * 1e3:8.0f} {fl / res["f_os"] / 1e9:5.1f} {res["f_old"] * 1e3:8.0f} | '
              f'{res["d_os"] * 1e3:8.0f} {res["d_old"] * 1e3:9.0f} | {res["wg"] * 1e3:7.0f} {fl / res["wg"] / 1e9:5.1f} {res["wg_old"] * 1e3:7.0f}')
    print(f'total: {tot["fl"] / 1e9:.1f} GFLOP per pass; forward os {tot["f_os"]:.2f} ms = {tot["fl"] / tot["f_os"] / 1e9:.1f} TF/s '
          f'({100 * tot["fl"] / tot["f_os"] / 1e9 / PEAK:.1f} % of {PEAK}), old {tot["f_old"]:.2f} ms; dgrad os {tot["d_os"]:.2f} ms, '
          f'old {tot["d_old"]:.2f} ms; wgrad {tot["wg"]:.2f} ms = {tot["fl"] / tot["wg"] / 1e9:.1f} TF/s, old {tot["wg_old"]:.2f} ms')


if __name__ == '__main__':
    main()
