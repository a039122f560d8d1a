"""Time the one-kernel tail of an encoder layer (csrc/layer_tail_x6.hip) beside the launch-per-product sequence it replaces,
at the bench frame's size.  python tools/tail_bench.py [M] -> one JSON line (us per call, algorithmic GB/s)."""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from sst_amd import dense as D  # noqa: E402


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main_bf16(m):
    from sst_amd import bf16 as B
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)

    def r(*s, sc=1.0):
        return (torch.randn(*s, generator=g) * sc).to(dev)
    o, x = r(m, 128).bfloat16(), r(m, 128).bfloat16()
    w_out, b_out, w1, b1, w2, b2 = r(128, 128, sc=.09), r(128, sc=.1), r(256, 128, sc=.09), r(256, sc=.1), r(128, 256, sc=.06), r(128, sc=.1)
    n1w, n1b, n2w, n2b = 1 + r(128, sc=.2), r(128, sc=.1), 1 + r(128, sc=.2), r(128, sc=.1)
    pos = (r(144, 128), torch.randint(0, 144, (m,), generator=g, dtype=torch.int32).to(dev))
    eps, act = 1e-5, 'gelu'
    keep = {}
    packed = B.tail_pack(w_out, w1, w2)
    t_pack = timed(lambda: B.tail_pack(w_out, w1, w2, out=packed))
    wo, w1s, w2s = B.shadow(w_out), B.shadow(w1), B.shadow(w2)
    w2t, w1t, wot = (B.shadow(w, transposed=True) for w in (w2, w1, w_out))

    def fused_fwd():
        keep['out'] = B.tail_fwd(o, x, packed, b_out, b1, b2, n1w, n1b, n2w, n2b, eps, act, pos=pos, out=keep.get('out'))

    def unfused_fwd():
        y1, s1, st1, _ = B.linear_add_ln(o, wo, b_out, x, n1w, n1b, eps)
        h, pre = B.tall_linear(y1, w1s, b1, B.EPI_GELU, want_pre=True)
        keep['u'] = B.linear_add_ln(h, w2s, b2, y1, n2w, n2b, eps, pos=pos)
    t_f, t_u = timed(fused_fwd), timed(unfused_fwd)
    out = keep['out']
    dy2 = r(m, 128).bfloat16()

    def fused_bwd():
        keep['b'] = B.tail_bwd(dy2, None, out['s2'], out['st2'], out['pre'], out['s1'], out['st1'], packed, n1w, n2w, act)

    def unfused_bwd():
        ds2, _, _ = B.add_ln_bwd(dy2, None, out['s2'], out['st2'], n2w)
        dpre = B.tall_linear(ds2, w2t, None, B.EPI_MUL_GELU_GRAD, aux_in=out['pre'])
        dy1 = B.tall_linear(dpre, w1t, None, B.EPI_ADD, aux_in=ds2)
        ds1, _, _ = B.add_ln_bwd(dy1, None, out['s1'], out['st1'], n1w)
        keep['ub'] = B.tall_linear(ds1, wot)
    tb_f, tb_u = timed(fused_bwd), timed(unfused_bwd)
    unit = m * 256 / 1e9
    print(json.dumps({'mode': 'bf16', 'm': m, 'pack_us': round(t_pack, 1),
                      'fwd_us': {'one_kernel': round(t_f, 1), 'launch_per_product': round(t_u, 1)},
                      'bwd_us': {'one_kernel': round(tb_f, 1), 'launch_per_product': round(tb_u, 1)},
                      'fwd_algorithmic_GBps': round(11 * unit / (t_f * 1e-6), 0), 'bwd_algorithmic_GBps': round(10 * unit / (tb_f * 1e-6), 0)}))


def main():
    if len(sys.argv) > 2 and sys.argv[2] == 'bf16':
        return main_bf16(int(sys.argv[1]))
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 90107
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)

    def r(*s, sc=1.0):
        return (torch.randn(*s, generator=g) * sc).to(dev)
    o, x = r(m, 128), r(m, 128)
    w_out, b_out, w1, b1, w2, b2 = r(128, 128, sc=.09), r(128, sc=.1), r(256, 128, sc=.09), r(256, sc=.1), r(128, 256, sc=.06), r(128, sc=.1)
    n1w, n1b, n2w, n2b = 1 + r(128, sc=.2), r(128, sc=.1), 1 + r(128, sc=.2), r(128, sc=.1)
    pos = (r(144, 128), torch.randint(0, 144, (m,), generator=g, dtype=torch.int32).to(dev))
    eps, act = 1e-5, 'gelu'
    D.set_matmul_mode('f32x6')
    keep = {}
    packed = D.encoder_tail_pack(w_out, w1, w2)
    t_pack = timed(lambda: D.encoder_tail_pack(w_out, w1, w2, out=packed))

    def fused_fwd():
        keep['out'] = D.encoder_tail_fwd(o, x, packed, b_out, b1, b2, n1w, n1b, n2w, n2b, eps, act, save=True, pos=pos,
                                         out=keep.get('out'))

    def unfused_fwd():
        y1, s1, st1, _ = D.lds_linear_add_ln(o, w_out, b_out, x, n1w, n1b, eps)
        h, pre = D.lds_linear(y1, w1, b1, D.EPI_GELU, want_pre=True)
        s2 = D.lds_linear(h, w2, b2, D.EPI_ADD, aux_in=y1)
        out = D.add_ln_fwd(s2, None, n2w, n2b, eps, pos=pos)
        keep['u'] = (y1, s1, st1, h, pre, s2, out)

    t_f, t_u = timed(fused_fwd), timed(unfused_fwd)
    out = keep['out']
    dy2 = r(m, 128)

    def fused_bwd():
        keep['b'] = D.encoder_tail_bwd(dy2, None, out['s2'], out['st2'], out['pre'], out['s1'], out['st1'], packed, n1w, n2w, act)

    def unfused_bwd():
        ds2, _, _ = D.add_ln_bwd(dy2, out['s2'], out['st2'], n2w)
        dpre = D.lds_linear(ds2, w2, None, D.EPI_MUL_GELU_GRAD, trans_w=True, aux_in=out['pre'])
        dy1 = D.lds_linear(dpre, w1, None, D.EPI_ADD, trans_w=True, aux_in=ds2, out=torch.empty_like(ds2))
        ds1, _, _ = D.add_ln_bwd(dy1, out['s1'], out['st1'], n1w)
        d_o = D.lds_linear(ds1, w_out, None, trans_w=True)
        keep['ub'] = (ds2, dpre, dy1, ds1, d_o)

    tb_f, tb_u = timed(fused_bwd), timed(unfused_bwd)
    unit = m * 512 / 1e9   # GB of one [M, 128] fp32 tensor
    print(json.dumps({
        'm': m, 'pack_us': round(t_pack, 1),
        'fwd_us': {'one_kernel': round(t_f, 1), 'launch_per_product': round(t_u, 1)},
        'bwd_us': {'one_kernel': round(tb_f, 1), 'launch_per_product': round(tb_u, 1)},
        'fwd_algorithmic_GBps': round(11 * unit / (t_f * 1e-6), 0),      # reads o, x; writes s1, y1, pre, h, s2, y2, y2p
        'bwd_algorithmic_GBps': round(10 * unit / (tb_f * 1e-6), 0),     # reads dy2, s2, pre, s1; writes ds2, dpre, ds1, d_o
    }))


if __name__ == '__main__':
    main()
