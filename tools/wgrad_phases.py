"""Phase breakdown of the wide weight-gradient kernel from in-kernel timestamps.
Build the instrumented library first (`make -C sst_amd/csrc libsst_amd_wgtiming.so`), then on the GPU:
    SST_AMD_LIB=$PWD/sst_amd/csrc/libsst_amd_wgtiming.so python tools/wgrad_phases.py
Prints, per shape, the median over workgroups of: entry -> address set-up done -> K loop done -> LDS reduction done
-> partials stored, the kernel's span (first entry to last store, all workgroups), and the shader clock during the
K loop (cycles / 100 MHz-clock time)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sst_amd import _lib  # noqa: E402
from sst_amd.dense import weight_bias_grad  # noqa: E402


def main():
    lib = _lib.load()
    fn = lib.sst_debug_wgrad_timestamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    fn.restype = ctypes.c_int
    m = int(os.environ.get('WG_M', 90107))
    ts = np.zeros((1024, 4, 8, 2), dtype=np.uint64)
    for out, inn in ((256, 128), (128, 128), (128, 256), (384, 128)):
        dy = torch.randn(m, out, device='cuda')
        x = torch.randn(m, inn, device='cuda')
        for _ in range(20):
            weight_bias_grad(dy, x, True)
        torch.cuda.synchronize()
        assert fn(ts.ctypes.data, ts.nbytes) == 0
        nwg = (out // 128) * (inn // 64) * max(8, (256 // ((out // 128) * (inn // 64))) & ~7)
        t = ts[:nwg].astype(np.int64)
        wall = t[..., 0] * 10.0 / 1000.0  # us (100 MHz)
        cyc = t[..., 1]
        w0 = wall[:, 0]  # wave 0 of every workgroup reaches all five points
        start = wall[:, :, 0].min()
        d = lambda a, b: float(np.median(w0[:, b] - w0[:, a]))
        loop_us = wall[:, :, 3] - wall[:, :, 1]
        loop_cyc = cyc[:, :, 3] - cyc[:, :, 1]
        ghz = float(np.median(loop_cyc / np.maximum(loop_us, 1e-3))) / 1000.0
        print('%dx%d (%d workgroups): setup %.2f us, K loop %.2f us (all waves: median %.2f, max %.2f), '
              'LDS reduce %.2f us, stores %.2f us; entry spread %.2f us, span %.2f us; shader clock in loop %.2f GHz'
              % (out, inn, nwg, d(0, 1), d(1, 3), float(np.median(loop_us)), float(loop_us.max()), d(3, 4), d(4, 5),
                 float(wall[:, :, 0].max() - start), float(w0[:, 5].max() - start), ghz), flush=True)


if __name__ == '__main__':
    main()
