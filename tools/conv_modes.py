#!/usr/bin/env python
"""Per-layer forward times of the sparse convolutions of a bench workload in the three multiply modes of the output-stationary
kernel: fp32 matrix pipe (csrc/spconv_os.hip), exact three-way bf16 split (csrc/spconv_os_x6.hip), two-way split
(csrc/spconv_os_x3.hip).  Usage: python tools/conv_modes.py [fsd|fsdv2] [points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as BW  # noqa: E402
from sst_amd import spconv as SP  # noqa: E402
from tools.conv_layers import timeit  # noqa: E402

DEV = torch.device('cuda:0')


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'fsd'
    spec = BW.WORKLOADS[what]
    n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else spec['points']
    torch.manual_seed(0)
    model = spec['cls']().to(DEV).train()
    clouds = [model.make_cloud(n_pts, 0, DEV)]
    layers, names = [], {}

    def post(mod, inp, out):
        datas = out.indice_dict.get(mod.indice_key) if (mod.indice_key is not None and not mod.conv1x1) else None
        rb = getattr(datas[2], '_sst_rulebook', None) if datas is not None else None
        if rb is not None:
            layers.append((mod, inp[0].features.detach(), rb))

    for name, m in model.named_modules():
        if isinstance(m, SP.SparseConvolution):
            m.register_forward_hook(post)
            names[m] = name
    with torch.no_grad():
        model(clouds)
    torch.cuda.synchronize()
    tot = {'f32': 0.0, 'f32x6': 0.0, 'f32x3': 0.0, 'fl': 0.0}
    print(f'{"layer":40s} {"rows":>7s} {"cin":>4s} {"cout":>4s} {"pairs":>8s} | {"f32 us":>7s} {"x6 us":>7s} {"x3 us":>7s} | x6 TF/s')
    for mod, x, rb in layers:
        w3 = mod.weight.detach().reshape(-1, mod.in_channels, mod.out_channels)
        fmap, frows = (rb.in2out, rb.n) if mod.inverse else (rb.out2in, rb.m)
        t = {}
        for mode in ('f32', 'f32x6', 'f32x3'):
            SP.set_conv_precision(mode)
            t[mode] = timeit(lambda: SP._gather_gemm(x, fmap, frows, w3, False, mod.out_channels, rb))
            tot[mode] += t[mode]
        SP.set_conv_precision('f32')
        fl = 2.0 * rb.total_pairs * mod.in_channels * mod.out_channels
        tot['fl'] += fl
        print(f'{names[mod][-40:]:40s} {frows:7d} {mod.in_channels:4d} {mod.out_channels:4d} {rb.total_pairs:8d} | '
              f'{t["f32"] * 1e3:7.0f} {t["f32x6"] * 1e3:7.0f} {t["f32x3"] * 1e3:7.0f} | {fl / t["f32x6"] / 1e9:6.1f}')
    print('total forward ms: ' + ', '.join(f'{m} {tot[m]:.2f}' for m in ('f32', 'f32x6', 'f32x3'))
          + f'; {tot["fl"] / 1e9:.1f} GFLOP: ' + ', '.join(f'{m} {tot["fl"] / tot[m] / 1e9:.1f} TF/s' for m in ('f32', 'f32x6', 'f32x3')))


if __name__ == '__main__':
    main()
