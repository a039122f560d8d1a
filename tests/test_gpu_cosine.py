"""GPU: scaled cosine attention INSIDE the attention kernels and the fused encoder chain.

Reference: CosineMultiheadAttention, mmdet3d/models/sst/cosine_msa.py:123-185 (normalize(q) normalize(k)^T / clamp(tau, tau_min),
softmax, v) - used by configs/sst_refactor/sst_waymoD5_1x_3class_centerhead.py:75 and configs/fsd/fsd_waymoD1_1x_sst_encoder.py:70.
The goldens produced by the reference's own module (sst_block_cosine*.npz) are checked through every path in
tests/test_gpu_sra.py::test_sst_block_matches_reference_golden; here: the kernels against a float64 restatement at the window sizes
of every tile class, the gradient of the scale, and that the fast paths (chain, layer executor) are the ones that run."""
import math

import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _plan(sizes, shuffle, seed):
    from sst_amd import kernels as K
    m = int(sum(sizes))
    g = torch.Generator().manual_seed(seed)
    tok = torch.randperm(m, generator=g) if shuffle else torch.arange(m)
    off = torch.zeros(len(sizes) + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(sizes), 0).int()
    plan = K.WindowPlan(tok.int().to(DEV), off.to(DEV), len(sizes), m, max(sizes), rows_in_window_order=not shuffle)
    return plan, tok.numpy(), off.numpy()


def _reference(qk, v, scale, tok, off, heads, go):
    """float64: per window and head softmax(normalize(q) normalize(k)^T * scale[h]) v, and its gradients by autograd"""
    qk = qk.double().requires_grad_(True)
    v = v.double().requires_grad_(True)
    scale = scale.double().requires_grad_(True)
    c = v.size(1)
    out = torch.zeros_like(v)
    for w in range(len(off) - 1):
        rows = torch.from_numpy(tok[off[w]:off[w + 1]]).long()
        q = torch.nn.functional.normalize(qk[rows, :c].reshape(len(rows), heads, 16), dim=2)
        k = torch.nn.functional.normalize(qk[rows, c:].reshape(len(rows), heads, 16), dim=2)
        s = torch.einsum('qhd,khd->hqk', q, k) * scale[:, None, None]
        o = torch.einsum('hqk,khd->qhd', torch.softmax(s, -1), v[rows].reshape(len(rows), heads, 16))
        out = out.index_put((rows,), o.reshape(len(rows), c))
    (out * go.double()).sum().backward()
    return out.detach(), qk.grad, v.grad, scale.grad


@pytest.mark.parametrize('heads', [8, 4])
@pytest.mark.parametrize('sizes,shuffle', [([1, 2, 15, 16, 17, 30], True), ([31, 47, 48, 60, 64], True), ([65, 80, 97, 100], True),
                                           ([113, 129, 144, 1], True), ([5, 33, 70, 144, 100, 60, 30], False)])
def test_cosine_kernels_match_float64(sizes, shuffle, heads):
    from sst_amd import kernels as K
    plan, tok, off = _plan(sizes, shuffle, seed=len(sizes) + heads)
    m, c = plan.n_tokens, heads * 16
    g = torch.Generator().manual_seed(m)
    qk = torch.randn(m, 2 * c, generator=g) * torch.rand(m, 1, generator=g).mul(3).add(0.1)     # rows of very different norms
    v = torch.randn(m, c, generator=g)
    go = torch.randn(m, c, generator=g)
    tau = torch.rand(heads, generator=g) * 0.5 + 0.05
    scale = 1.0 / tau
    assert K.cosine_kernels_ok(plan, heads)
    qk_d, v_d = qk.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    sc_d = scale.to(DEV).requires_grad_(True)
    out = K.sra_cosine_attention_qk_v(qk_d, v_d, sc_d, plan, heads)
    out.backward(go.to(DEV))
    want, dqk, dv, dsc = _reference(qk, v, scale, tok, off, heads, go)

    def close(got, ref, tol, what):
        err = float((got.detach().cpu().double() - ref).abs().max())
        assert err <= tol * max(1.0, float(ref.abs().max())), (what, err, float(ref.abs().max()))
    close(out, want, 2e-6, 'output')
    close(v_d.grad, dv, 5e-6, 'dv')
    close(qk_d.grad, dqk, 2e-5, 'dq | dk')
    close(sc_d.grad, dsc, 2e-5, 'd head_scale')
    # the gradient of an un-normalised row is orthogonal to the row (what the projection inside the kernel enforces)
    q, dq = qk_d.detach()[:, :c].reshape(m, heads, 16), qk_d.grad[:, :c].reshape(m, heads, 16)
    assert float(((q * dq).sum(-1).abs() / (q.norm(dim=-1) * dq.norm(dim=-1) + 1e-20)).max()) < 1e-4


def _cosine_backbone(blocks, layer_cfg, seed=5):
    import sst_amd
    torch.manual_seed(seed)
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128] * blocks, nhead=[8] * blocks, num_blocks=blocks,
                                      dim_feedforward=[256] * blocks, output_shape=[468, 468], num_attached_conv=0, to_bev=False,
                                      debug=False, layer_cfg=layer_cfg)).to(DEV).train()
    if not layer_cfg.get('cosine', False):
        return net
    with torch.no_grad():           # temperatures away from their initial 1 (and one below tau_min: the clamp must cut its gradient)
        for i, blk in enumerate(net.block_list):
            for j, enc in enumerate(blk.encoder_list):
                enc.win_attn.self_attn.tau.fill_(0.3 + 0.2 * j).mul_(1.0 if (i, j) != (0, 1) else 0.01)
    return net


def _frame(n_voxels, seed):
    g = torch.Generator().manual_seed(seed)
    side = int(math.ceil(math.sqrt(n_voxels * 2.2)))
    cells = torch.randperm(side * side, generator=g)[:n_voxels].sort()[0]
    coors = torch.stack([torch.zeros_like(cells), torch.zeros_like(cells), cells // side + 20, cells % side + 20], 1).to(DEV)
    return coors, torch.randn(n_voxels, 128, generator=g).to(DEV), torch.randn(n_voxels, 128, generator=g).to(DEV)


def _step(net, layer, feats0, coors, up):
    net.zero_grad(set_to_none=True)
    feats = feats0.clone().requires_grad_(True)
    out = net(layer(feats, coors, 1))[0]['voxel_feats']
    (out * up[:out.size(0)]).sum().backward()
    return out.detach().clone(), feats.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('layer_cfg', [dict(cosine=True, tau_min=0.01), dict(cosine=True, tau_min=0.01, non_shared_tau=True)])
@pytest.mark.parametrize('n_voxels', [2500, 12000])
def test_cosine_layers_run_in_the_fused_chain_and_match_the_normalise_outside_path(layer_cfg, n_voxels, monkeypatch):
    """cosine encoder layers take the encoder chain (12 000 voxels: the one-call layer executor) - the per-layer path is never
    entered - and give what the round-4 arithmetic gave (torch F.normalize / division around the standard kernel)"""
    import sst_amd
    from sst_amd import kernels as K
    from sst_amd import sst_basic_block as SB
    coors, feats0, up = _frame(n_voxels, 3)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True).eval()
    net = _cosine_backbone(2, layer_cfg)
    execs = []
    orig_exec = SB._layer_exec_fwd
    monkeypatch.setattr(SB, '_layer_exec_fwd', lambda *a, **k: (execs.append(1), orig_exec(*a, **k))[1])
    with monkeypatch.context() as mp:
        mp.setattr(SB.BasicShiftBlockV2, 'forward', lambda *a, **k: (_ for _ in ()).throw(AssertionError('per-layer path')))
        fast = _step(net, layer, feats0, coors, up)
    assert (len(execs) == 4) == (n_voxels >= 4096), 'the layer executor takes cosine layers from 4 096 tokens on'
    # the round-4 path: modular layers, q / k normalised and divided by tau in torch around the standard kernel
    monkeypatch.setattr(K, 'cosine_kernels_ok', lambda *a, **k: False)
    slow = _step(net, layer, feats0, coors, up)
    assert float((fast[0] - slow[0]).abs().max()) <= 4e-5      # two fp32 evaluations of 4 layers: 2.3e-5 measured (round 6: one-kernel tail)
    assert float((fast[1] - slow[1]).abs().max()) <= 5e-5 * max(1.0, float(slow[1].abs().max()))      # 2.0e-5 measured
    assert fast[2].keys() == slow[2].keys() and any(n.endswith('tau') for n in fast[2])
    for n in fast[2]:
        sc = max(1.0, float(slow[2][n].abs().max()))
        assert float((fast[2][n] - slow[2][n]).abs().max()) <= 1e-4 * sc, n
    clamped = fast[2]['block_list.0.encoder_list.1.win_attn.self_attn.tau']
    assert float(clamped.abs().max()) == 0.0, 'tau below tau_min: the clamp passes no gradient'
    assert float(fast[2]['block_list.0.encoder_list.0.win_attn.self_attn.tau'].abs().max()) > 0.0


def test_cosine_layer_executor_equals_the_python_sequence():
    import sst_amd
    from sst_amd import sst_basic_block as SB
    coors, feats0, up = _frame(9000, 4)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True).eval()
    net = _cosine_backbone(2, dict(cosine=True, tau_min=0.01))
    try:
        SB._LAYER_EXEC = 1
        a = _step(net, layer, feats0, coors, up)
        SB._LAYER_EXEC = 0
        b = _step(net, layer, feats0, coors, up)
    finally:
        SB._LAYER_EXEC = 1
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n in a[2]:
        assert torch.equal(a[2][n], b[2][n]), n


def test_independent_xp_keeps_its_own_gradient_in_the_split_mode():
    """an ``xp`` that is NOT x + constant handed to FusedEncoderLayerFn must get its own gradient in the
    exact-split mode too (the fold of d(xp) into d(x) is only for callers that declare xp_shares_x)"""
    from sst_amd import dense as D
    from sst_amd.sst_basic_block import FusedEncoderLayerFn
    import sst_amd
    coors, feats0, up = _frame(6000, 9)
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, debug=False,
                                    mute=True).eval()
    info = layer(feats0, coors, 1)
    plan = info['sra_plan_shift0']
    net = _cosine_backbone(1, dict())
    enc = net.block_list[0].encoder_list[0]
    attn = enc.win_attn.self_attn
    res = {}
    xp0 = torch.randn_like(info['voxel_feats'])                      # independent of x
    for mode in ('f32', 'f32x6'):
        with D.matmul_mode_scope(mode):
            x = info['voxel_feats'].clone().requires_grad_(True)
            xp = xp0.clone().requires_grad_(True)
            out = FusedEncoderLayerFn.apply(x, None, plan, 8, 0, 'gelu', attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight,
                                            attn.out_proj.bias, enc.linear1.weight, enc.linear1.bias, enc.linear2.weight,
                                            enc.linear2.bias, enc.norm1.weight, enc.norm1.bias, enc.norm2.weight, enc.norm2.bias,
                                            enc.norm1.eps, xp, None)
            (out * up[:out.size(0)]).sum().backward()
            res[mode] = (x.grad.clone(), xp.grad.clone())
    assert res['f32x6'][1] is not None and float(res['f32x6'][1].abs().max()) > 0
    for a, b in zip(res['f32'], res['f32x6']):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
