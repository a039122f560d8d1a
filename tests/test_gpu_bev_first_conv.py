"""recover_bev + the first attached convolution without the dense input canvas
(sst_amd/backbones.py: sparse_first_conv) against the dense path (recover_bev -> nn.Conv2d), sst_v2.py:139-197."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _voxels(rng, m, batch, ny, nx):
    cells = rng.choice(batch * ny * nx, size=m, replace=False)
    b, rest = cells // (ny * nx), cells % (ny * nx)
    return torch.from_numpy(np.stack([b, np.zeros_like(b), rest // nx, rest % nx], 1)).long()


@pytest.mark.parametrize('k,d', [(3, 2), (3, 1), (5, 1)])
@pytest.mark.parametrize('c_in,c_out', [(128, 128), (64, 96)])
def test_sparse_first_conv_equals_conv_of_the_canvas(k, d, c_in, c_out):
    from sst_amd.backbones import recover_bev, sparse_first_conv
    rng = np.random.default_rng(k * 10 + d + c_in)
    batch, ny, nx, m = 2, 60, 52, 1900
    coors = _voxels(rng, m, batch, ny, nx).to(DEV)
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(c_in, c_out, k, dilation=d, padding=d * (k - 1) // 2, bias=False).to(DEV)
    feats = torch.randn(m, c_in, device=DEV)
    gout = torch.randn(batch, c_out, ny, nx, device=DEV)
    fa, fb = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
    y = sparse_first_conv(fa, coors, batch, (ny, nx), conv)
    (y * gout).sum().backward()
    gw = conv.weight.grad.clone()
    conv.zero_grad()
    y_ref = conv(recover_bev(fb, coors, batch, (ny, nx)))
    (y_ref * gout).sum().backward()
    assert y.shape == y_ref.shape
    scale = max(1.0, float(y_ref.abs().max()))
    assert float((y - y_ref).abs().max()) < 1e-4 * scale
    assert float((fa.grad - fb.grad).abs().max()) < 1e-4 * max(1.0, float(fb.grad.abs().max()))
    assert float((gw - conv.weight.grad).abs().max()) < 1e-4 * max(1.0, float(conv.weight.grad.abs().max()))


def test_sstv2_bev_head_takes_the_sparse_first_conv(monkeypatch):
    """the whole BEV head (3 x conv 3x3 dilation 2 + batch norm + ReLU) through SSTv2.bev_and_attached_convs with and
    without the sparse first convolution: same output, same gradients"""
    import sst_amd
    rng = np.random.default_rng(3)
    batch, ny, nx, m, c = 2, 48, 56, 1500, 128
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[c], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                      output_shape=[ny, nx], num_attached_conv=3, conv_in_channel=c, conv_out_channel=c,
                                      norm_cfg=dict(type='BN2d', eps=1e-3, momentum=0.01), debug=True)).to(DEV).train()
    coors = _voxels(rng, m, batch, ny, nx).to(DEV)
    feats = torch.randn(m, c, device=DEV)
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('SST_BEV_SPARSE_FIRST_CONV', flag)
        net.zero_grad()
        f = feats.clone().requires_grad_(True)
        y = net.bev_and_attached_convs(f, coors, batch)
        y.square().mean().backward()
        outs.append((y.detach(), f.grad.clone(), net.conv_layer[0][0].weight.grad.clone(), net.conv_layer[0][1].weight.grad.clone()))
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max()))
