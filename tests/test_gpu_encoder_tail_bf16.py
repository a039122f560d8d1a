"""GPU: the one-kernel tail of an encoder layer in the reduced-precision mode (csrc/layer_tail_bf16.hip,
sst_encoder_tail_{fwd,bwd}_bf16: bf16 storage, fp32 accumulation) against the launch-per-product bf16 kernels it replaces
(csrc/dense_bf16.hip, csrc/dense.hip add_ln_*_bf16) - same roundings at the same places, so the outputs agree to a bf16 ulp of
their scale - and against the fp32 chain in float64 within the bf16 tolerance of tests/test_gpu_bf16.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF = torch.bfloat16


def _params(seed):
    g = torch.Generator().manual_seed(seed)

    def r(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).to(DEV)
    return dict(w_out=r(128, 128, s=0.09), b_out=r(128, s=0.1), w1=r(256, 128, s=0.09), b1=r(256, s=0.1),
                w2=r(128, 256, s=0.06), b2=r(128, s=0.1), n1w=1 + r(128, s=0.2), n1b=r(128, s=0.1),
                n2w=1 + r(128, s=0.2), n2b=r(128, s=0.1))


def _unfused(o, x, p, act, eps, pos):
    from sst_amd import bf16 as B
    wo, w1, w2 = (B.shadow(p[k]) for k in ('w_out', 'w1', 'w2'))
    y1, s1, st1, _ = B.linear_add_ln(o, wo, p['b_out'], x, p['n1w'], p['n1b'], eps)
    h, pre = B.tall_linear(y1, w1, p['b1'], B.EPI_GELU if act == 'gelu' else B.EPI_RELU, want_pre=True)
    y2, s2, st2, y2p = B.linear_add_ln(h, w2, p['b2'], y1, p['n2w'], p['n2b'], eps, pos=pos)
    return dict(s1=s1, st1=st1, y1=y1, pre=pre, h=h, s2=s2, st2=st2, y2=y2, y2p=y2p)


def _close(a, b, ulps=2.0):
    """within `ulps` bf16 steps of the tensor's scale (2^-8 relative per step)"""
    a, b = a.float(), b.float()
    scale = max(1e-6, float(b.abs().max()))
    return float((a - b).abs().max()) <= ulps * scale * 2.0 ** -8


@pytest.mark.parametrize('m,act,with_pos', [(1, 'gelu', False), (17, 'relu', True), (129, 'gelu', True), (4099, 'relu', False),
                                            (20011, 'gelu', True)])
def test_bf16_tail_matches_the_launch_per_product_kernels(m, act, with_pos):
    from sst_amd import bf16 as B
    eps = 1e-5
    g = torch.Generator().manual_seed(m)
    o, x = torch.randn(m, 128, generator=g).to(DEV).to(BF), torch.randn(m, 128, generator=g).to(DEV).to(BF)
    p = _params(3)
    pos = None
    if with_pos:
        pos = (torch.randn(144, 128, generator=g).to(DEV), torch.randint(0, 144, (m,), generator=g, dtype=torch.int32).to(DEV))
    assert B.tail_ok(o, x, p['w_out'], p['w1'], p['w2'])
    packed = B.tail_pack(p['w_out'], p['w1'], p['w2'])
    out = B.tail_fwd(o, x, packed, p['b_out'], p['b1'], p['b2'], p['n1w'], p['n1b'], p['n2w'], p['n2b'], eps, act, pos=pos)
    ref = _unfused(o, x, p, act, eps, pos)
    for k in ('s1', 'y1', 'pre', 'h', 's2', 'y2') + (('y2p',) if with_pos else ()):
        assert _close(out[k], ref[k]), k
    for k in ('st1', 'st2'):
        assert float((out[k] - ref[k]).abs().max() / ref[k].abs().max()) < 2e-2, k

    dy2 = torch.randn(m, 128, generator=g).to(DEV).to(BF)
    dy2p = torch.randn(m, 128, generator=g).to(DEV).to(BF) if with_pos else None
    ds2, dpre, ds1, d_o, dn = B.tail_bwd(dy2, dy2p, out['s2'], out['st2'], out['pre'], out['s1'], out['st1'], packed, p['n1w'],
                                         p['n2w'], act)
    # the launch-per-product backward on the SAME saved tensors
    rs2, rn2w, rn2b = B.add_ln_bwd(dy2, dy2p, out['s2'], out['st2'], p['n2w'])
    w2t, w1t, wot = (B.shadow(p[k], transposed=True) for k in ('w2', 'w1', 'w_out'))
    rpre = B.tall_linear(rs2, w2t, None, B.EPI_MUL_GELU_GRAD if act == 'gelu' else B.EPI_MUL_RELU_GRAD, aux_in=out['pre'])
    rdy1 = B.tall_linear(rpre, w1t, None, B.EPI_ADD, aux_in=rs2)
    rs1, rn1w, rn1b = B.add_ln_bwd(rdy1, None, out['s1'], out['st1'], p['n1w'])
    rdo = B.tall_linear(rs1, wot)
    assert _close(ds2, rs2) and _close(dpre, rpre, 3.0) and _close(ds1, rs1, 4.0) and _close(d_o, rdo, 4.0)
    for got, want in ((dn[0], rn2w), (dn[1], rn2b), (dn[2], rn1w), (dn[3], rn1b)):
        sc = max(1e-6, float(want.abs().max()))
        assert float((got - want).abs().max()) <= 2e-2 * sc


def test_bf16_tail_is_deterministic_at_90k_tokens():
    from sst_amd import bf16 as B
    m, act, eps = 90107, 'gelu', 1e-5
    g = torch.Generator().manual_seed(9)
    o, x, dy2 = (torch.randn(m, 128, generator=g).to(DEV).to(BF) for _ in range(3))
    p = _params(11)
    ref = None
    for _ in range(4):
        packed = B.tail_pack(p['w_out'], p['w1'], p['w2'])
        out = B.tail_fwd(o, x, packed, p['b_out'], p['b1'], p['b2'], p['n1w'], p['n1b'], p['n2w'], p['n2b'], eps, act)
        bwd = B.tail_bwd(dy2, None, out['s2'], out['st2'], out['pre'], out['s1'], out['st1'], packed, p['n1w'], p['n2w'], act)
        cur = [out[k].clone() for k in ('s1', 'y1', 'pre', 'h', 's2', 'y2')] + [t.clone() for t in bwd]
        if ref is None:
            ref = cur
        else:
            for a, b in zip(cur, ref):
                assert torch.equal(a, b)
    # against float64 on the last rows (a partial workgroup): bf16 tolerance
    d = {k: v.double() for k, v in p.items()}
    o64, x64 = o[-300:].double(), x[-300:].double()
    y1 = torch.nn.functional.layer_norm(x64 + o64 @ d['w_out'].t() + d['b_out'], (128,), d['n1w'], d['n1b'], eps)
    s2 = y1 + torch.nn.functional.gelu(y1 @ d['w1'].t() + d['b1']) @ d['w2'].t() + d['b2']
    y2 = torch.nn.functional.layer_norm(s2, (128,), d['n2w'], d['n2b'], eps)
    assert float((ref[5][-300:].double() - y2).abs().max()) < 8e-2
