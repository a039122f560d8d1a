#!/usr/bin/env python
"""Developer probe: csrc/tall_gemm.hip against the library GEMM on the encoder-layer shapes (time + max error)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sst_amd.dense import tall_gemm  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=20, warmup=3):
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


m = int(sys.argv[1]) if len(sys.argv) > 1 else 90107
torch.manual_seed(0)
for (k, trans, acc) in ((128, False, False), (256, False, False), (128, True, False), (256, True, True), (128, True, True)):
    n = 128
    x = torch.randn(m, k, device=dev)
    w = torch.randn(k, n, device=dev) if trans else torch.randn(n, k, device=dev)
    b = torch.randn(n, device=dev)
    y0 = torch.randn(m, n, device=dev)
    wt = w if trans else w.t()
    ref = x.double() @ wt.double() + b.double() + (y0.double() if acc else 0)
    out = y0.clone() if acc else None
    got = tall_gemm(x, w, b, trans_w=trans, out=out, accumulate=acc)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    lib_out = torch.empty(m, n, device=dev)
    t_lib = timeit(lambda: torch.addmm(b, x, wt, out=lib_out))
    buf = y0.clone()
    t_own = timeit(lambda: tall_gemm(x, w, b, trans_w=trans, out=buf, accumulate=acc))
    fl = 2.0 * m * n * k
    print(f'm={m} n={n} k={k} trans={int(trans)} acc={int(acc)}: rel.err {err:.1e}  library {t_lib * 1e3:6.1f} us '
          f'({fl / t_lib / 1e9:5.1f} TF/s)   tall_gemm {t_own * 1e3:6.1f} us ({fl / t_own / 1e9:5.1f} TF/s)')
