"""Microbenchmark of the LDS-resident linears in the three multiply modes (csrc/dense_f32.hip, dense_f32x3.hip, dense_f32x6.hip)
at the headline size (90 107 tokens): us per launch by HIP events over 50 launches, per shape and epilogue, and the q | k | v
single launch of the f32x6 mode against its two separate launches.  python tools/dense_x6_bench.py [M]"""
import sys

import torch

sys.path.insert(0, '.')
from sst_amd import dense as D  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 90107
dev = 'cuda:0'


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator().manual_seed(0)
rows = []
for k, n in ((128, 128), (128, 256), (256, 128)):
    x = torch.randn(M, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.1).to(dev)
    wt = (torch.randn(k, n, generator=g) * 0.1).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    aux = torch.randn(M, n, generator=g).to(dev)
    res = torch.randn(M, 128, generator=g).to(dev)
    lw, lb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    for mode in ('f32', 'f32x3', 'f32x6'):
        D.set_matmul_mode(mode)
        r = {'shape': (k, n), 'mode': mode}
        r['bias'] = timeit(lambda: D.lds_linear(x, w, b))
        r['gelu'] = timeit(lambda: D.lds_linear(x, w, b, D.EPI_GELU, want_pre=True))
        r['dgrad'] = timeit(lambda: D.lds_linear(x, wt, None, trans_w=True))
        r['dgrad*gelu\''] = timeit(lambda: D.lds_linear(x, wt, None, D.EPI_MUL_GELU_GRAD, trans_w=True, aux_in=aux))
        r['add'] = timeit(lambda: D.lds_linear(x, w, b, D.EPI_ADD, aux_in=aux))
        if n == 128 and D.lds_linear_add_ln_ok(x, w, res, 128):
            r['+LN'] = timeit(lambda: D.lds_linear_add_ln(x, w, b, res, lw, lb, 1e-5))
        rows.append(r)
        print(r, flush=True)
D.set_matmul_mode('f32x6')
x = torch.randn(M, 128, generator=g).to(dev)
xp = torch.randn(M, 128, generator=g).to(dev)
w = (torch.randn(384, 128, generator=g) * 0.1).to(dev)
b = torch.randn(384, generator=g).to(dev)
print({'qkv one launch': timeit(lambda: D.lds_linear_qkv(xp, x, w, b)),
       'qk + v': timeit(lambda: (D.lds_linear(xp, w[:256], b[:256]), D.lds_linear(x, w[256:], b[256:])))})
# the five weight gradients of a layer, as the backward pass groups them
dqkv = torch.randn(M, 384, generator=g).to(dev)
ds2, ds1 = torch.randn(M, 128, generator=g).to(dev), torch.randn(M, 128, generator=g).to(dev)
dpre, h = torch.randn(M, 256, generator=g).to(dev), torch.randn(M, 256, generator=g).to(dev)
y1, o = torch.randn(M, 128, generator=g).to(dev), torch.randn(M, 128, generator=g).to(dev)
grp1 = [(ds2, h, torch.empty(128, 256, device=dev), torch.empty(128, device=dev)),
        (dpre, y1, torch.empty(256, 128, device=dev), torch.empty(256, device=dev))]
grp2 = [(ds1, o, torch.empty(128, 128, device=dev), torch.empty(128, device=dev)),
        (dqkv[:, :256], xp, torch.empty(256, 128, device=dev), torch.empty(256, device=dev)),
        (dqkv[:, 256:], x, torch.empty(128, 128, device=dev), torch.empty(128, device=dev))]
for mode in ('f32', 'f32x6'):
    D.set_matmul_mode(mode)
    print({'wgrad mode': mode, 'group1 (dW2, dW1)': timeit(lambda: D.weight_bias_grad_group(grp1)),
           'group2 (dWo, dWqk, dWv)': timeit(lambda: D.weight_bias_grad_group(grp2))})
res = torch.randn(M, 128, generator=g).to(dev)
lw, lb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
print({'add_ln_fwd pass (fp32)': timeit(lambda: D.add_ln_fwd(res, None, lw, lb, 1e-5))})
D.set_matmul_mode('f32')
