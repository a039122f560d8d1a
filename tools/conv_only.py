#!/usr/bin/env python
"""One sparse convolution of the FSD / FSDv2 bench pass in a loop (for `rocprofv3 --pmc` / `--stats`): the layers are
captured as in tools/conv_layers.py, then the forward contraction of layer <index> runs <iters> times.
Usage: python tools/conv_only.py [fsd|fsdv2] <layer index> [iters] [tile_cfg]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as BW  # noqa: E402
from sst_amd import spconv as SP  # noqa: E402

DEV = torch.device('cuda:0')


def capture(what):
    spec = BW.WORKLOADS[what]
    torch.manual_seed(0)
    model = spec['cls']().to(DEV).train()
    clouds = [model.make_cloud(spec['points'], 0, DEV)]
    layers, names = [], {}

    def post(mod, inp, out):
        if mod.conv1x1:
            return
        datas = out.indice_dict.get(mod.indice_key) if mod.indice_key is not None else None
        rb = getattr(datas[2], '_sst_rulebook', None) if datas is not None else None
        if rb is not None:
            layers.append(dict(mod=mod, x=inp[0].features.detach(), rb=rb, name=names[mod]))

    for name, m in model.named_modules():
        if isinstance(m, SP.SparseConvolution):
            m.register_forward_hook(post)
            names[m] = name
    with torch.no_grad():
        model(clouds)
    torch.cuda.synchronize()
    return layers


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'fsd'
    index = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    L = capture(what)[index]
    mod, x, rb = L['mod'], L['x'], L['rb']
    w3 = mod.weight.detach().reshape(-1, mod.in_channels, mod.out_channels)
    fmap, frows = (rb.in2out, rb.n) if mod.inverse else (rb.out2in, rb.m)
    for _ in range(iters):
        SP._gather_gemm(x, fmap, frows, w3, False, mod.out_channels, rb, tile_cfg=tile)
    torch.cuda.synchronize()
    print(L['name'], 'rows', frows, 'cin', mod.in_channels, 'cout', mod.out_channels, 'pairs', rb.total_pairs)


if __name__ == '__main__':
    main()
