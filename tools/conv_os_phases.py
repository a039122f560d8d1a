#!/usr/bin/env python
"""Phase breakdown of the output-stationary sparse convolution (csrc/spconv_os.hip) from in-kernel clocks.
Build the instrumented library first (`make -C sst_amd/csrc libsst_amd_ostiming.so`), then on the GPU:
    SST_AMD_LIB=$PWD/sst_amd/csrc/libsst_amd_ostiming.so python tools/conv_os_phases.py [fsd|fsdv2] [layer indices ...]
Per layer: the launch as the host sees it, and over the waves of the first 4096 units the set-up (index image, live
masks), the first fill, the loop split into {issue of the next stage's loads, MFMA issue, wait for the loads + LDS
writes, barrier}, the stores; staged and multiplied stages per wave; the span from the first entry to the last exit."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv_only import capture  # noqa: E402
from sst_amd import _lib  # noqa: E402
from sst_amd import spconv as SP  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'fsd'
    sel = [int(a) for a in sys.argv[2:]] or [0, 3, 6]
    tile = int(os.environ.get('OS_TILE', 0))
    lib = _lib.load()
    fn = lib.sst_debug_conv_os_timestamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    fn.restype = ctypes.c_int
    layers = capture(what)
    ts = np.zeros((4096, 4, 16), dtype=np.uint64)
    for i in sel:
        L = layers[i]
        mod, x, rb = L['mod'], L['x'], L['rb']
        w3 = mod.weight.detach().reshape(-1, mod.in_channels, mod.out_channels)
        fmap, frows = (rb.in2out, rb.n) if mod.inverse else (rb.out2in, rb.m)
        for _ in range(3):
            SP._gather_gemm(x, fmap, frows, w3, False, mod.out_channels, rb, tile_cfg=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        SP._gather_gemm(x, fmap, frows, w3, False, mod.out_channels, rb, tile_cfg=tile)
        e1.record()
        torch.cuda.synchronize()
        assert fn(ts.ctypes.data, ts.nbytes) == 0
        rows = lib.sst_spconv_conv_os_tile_rows(frows, mod.out_channels, tile)
        units = min(4096, -(-frows // rows) * -(-mod.out_channels // 64))
        t = ts[:units].astype(np.int64)
        us = lambda a: a * 0.01
        mark = us(t[:, :, :5])
        acc = us(t[:, :, 8:12])
        n_st, n_live = t[:, :, 12], t[:, :, 13]
        med = lambda a: float(np.median(a))
        print(f'{L["name"][-44:]:44s} rows {frows} x {rows}-row tiles, {units} units sampled; launch {e0.elapsed_time(e1) * 1e3:.0f} us '
              f'(incl. weight packing), span {mark[:, :, 4].max() - mark[:, :, 0].min():.0f} us')
        print(f'   per wave (median / mean): set-up {med(mark[:, :, 1] - mark[:, :, 0]):.2f} / {(mark[:, :, 1] - mark[:, :, 0]).mean():.2f} us, '
              f'first fill {med(mark[:, :, 2] - mark[:, :, 1]):.2f} / {(mark[:, :, 2] - mark[:, :, 1]).mean():.2f}, '
              f'loop {med(mark[:, :, 3] - mark[:, :, 2]):.2f} / {(mark[:, :, 3] - mark[:, :, 2]).mean():.2f}, '
              f'stores {med(mark[:, :, 4] - mark[:, :, 3]):.2f} / {(mark[:, :, 4] - mark[:, :, 3]).mean():.2f}, '
              f'whole {med(mark[:, :, 4] - mark[:, :, 0]):.2f} / {(mark[:, :, 4] - mark[:, :, 0]).mean():.2f}')
        st = np.maximum(n_st, 1)
        print(f'   stages per wave {n_st.mean():.1f} staged, {n_live.mean():.1f} multiplied; per staged stage (mean us): '
              f'load issue {(acc[:, :, 0] / st).mean():.2f}, MFMA issue {(acc[:, :, 1] / st).mean():.2f} '
              f'({(acc[:, :, 1].sum() / max(1, n_live.sum())):.2f} per multiplied stage), loads + LDS writes {(acc[:, :, 2] / st).mean():.2f}, '
              f'barrier {(acc[:, :, 3] / st).mean():.2f}')
        # how many waves of a CU-sized group are resident over time: entry / exit histogram in 10 us bins
        t0 = mark[:, :, 0].min()
        ent, ex = mark[:, 0, 0] - t0, mark[:, 0, 4] - t0
        edges = np.arange(0, ex.max() + 10, 10)
        resident = [(int(((ent <= e) & (ex > e)).sum())) for e in edges]
        print('   sampled workgroups resident every 10 us:', resident)


if __name__ == '__main__':
    main()
