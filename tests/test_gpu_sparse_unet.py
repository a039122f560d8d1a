"""GPU: SimpleSparseUNet (FSD's segmentor backbone) on sst_amd.spconv against outputs and gradients of the reference's
own SimpleSparseUNet / sparse blocks / vendored spconv Python executed on CPU (tests/golden/sparse_unet.npz, generated
by tests/golden/make_golden.py::gen_sparse_unet through oracle/ref_loader.load_reference_spconv).  fp32 on both
sides: 1e-3 of the tensor's scale (the strided layers number their outputs differently, sums run in another order)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

DEV = 'cuda:0'
CFG = dict(in_channels=8, sparse_shape=[16, 40, 40], order=('conv', 'norm', 'act'),
           norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=32,
           encoder_channels=((16, ), (16, 16, 16), (32, 32, 32), (32, 32, 32)),
           encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
           decoder_channels=((32, 32, 32), (32, 32, 16), (16, 16, 16), (16, 16, 16)),
           decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)))


def _state(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g if k.startswith('w::')}


def test_state_dict_keys_and_shapes_match_the_reference_module():
    """CPU: the mirror builds the parameter / buffer tree of the reference class (names and shapes)"""
    import sst_amd
    g = load_golden('sparse_unet.npz')
    net = sst_amd.BACKBONES.build(dict(type='SimpleSparseUNet', **CFG))
    mine, ref = net.state_dict(), _state(g)
    assert set(mine) == set(ref)
    assert all(tuple(mine[k].shape) == tuple(ref[k].shape) for k in ref)
    net.load_state_dict(ref, strict=True)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_simple_sparse_unet_matches_reference_golden(mode):
    import sst_amd
    g = load_golden('sparse_unet.npz')
    net = sst_amd.SimpleSparseUNet(**CFG)
    net.load_state_dict(_state(g), strict=True)
    net = net.to(DEV).train(mode == 'train')
    x = torch.from_numpy(g['in::features']).to(DEV).requires_grad_(True)
    ind = torch.from_numpy(g['in::indices']).to(DEV)
    out = net({'voxel_feats': x, 'voxel_coors': ind})[0]
    assert torch.equal(out['voxel_coors'], ind) and list(out['sparse_shape']) == CFG['sparse_shape']
    assert out['batch_size'] == 2 and out['decoder_features'] == []
    (out['voxel_feats'] * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()

    def close(got, want, what):
        want = torch.from_numpy(want)
        err = float((got.detach().cpu() - want).abs().max())
        assert err < 1e-3 * max(1.0, float(want.abs().max())), (what, err)

    close(out['voxel_feats'], g[f'out::{mode}::voxel_feats'], 'voxel_feats')
    close(x.grad, g[f'out::{mode}::grad_features'], 'grad_features')
    params = dict(net.named_parameters())
    for key in g:
        if key.startswith(f'out::{mode}::grad::'):
            name = key.split('::', 3)[3]
            close(params[name].grad, g[key], name)


@pytest.mark.gpu
def test_fsd_config_backbone_runs_at_scale():
    """the SimpleSparseUNet of configs/fsd/fsd_waymoD1_1x.py:39-51 (64-channel stem, five stages up to 256
    channels) on 60 k voxels: shapes, finiteness, one backward pass, rulebook sharing through indice keys."""
    import sst_amd
    net = sst_amd.BACKBONES.build(dict(
        type='SimpleSparseUNet', in_channels=64, sparse_shape=[32, 640, 640], order=('conv', 'norm', 'act'),
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), base_channels=64, output_channels=128,
        encoder_channels=((64, ), (64, 64, 64), (64, 64, 64), (128, 128, 128), (256, 256, 256)),
        encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1), (1, 1, 1)),
        decoder_channels=((256, 256, 128), (128, 128, 64), (64, 64, 64), (64, 64, 64), (64, 64, 64)),
        decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1)))).to(DEV)
    rng = np.random.default_rng(2)
    hs = [16, 320, 320]
    vol = int(np.prod(hs))
    lin = rng.choice(2 * vol, 15000, replace=False)
    b, r = lin // vol, lin % vol
    base = np.stack([b, r // (hs[1] * hs[2]), (r // hs[2]) % hs[1], r % hs[2]], 1)
    ind = np.unique(np.concatenate([base * [1, 2, 2, 2] + [0, 0, dy, dx] for dy in (0, 1) for dx in (0, 1)]), axis=0)
    ind = torch.from_numpy(ind.astype(np.int32)).to(DEV)
    x = torch.randn(ind.size(0), 64, device=DEV, requires_grad=True)
    out = net({'voxel_feats': x, 'voxel_coors': ind})[0]
    assert out['voxel_feats'].shape == (ind.size(0), 64) and torch.isfinite(out['voxel_feats']).all()
    out['voxel_feats'].square().mean().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


MIXER_CFG = dict(in_channels=8, sparse_shape=[16, 40, 40], order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=24,
                 encoder_channels=((16, ), (16, 16), (16, 16)), encoder_paddings=((1, ), (1, 1), (1, 1)),
                 decoder_channels=((16, 16, 16), (16, 16, 16), (16, 16, 16)),
                 decoder_paddings=((1, 1), (1, 1), (1, 1)))


def test_voxel_mixer_state_dict_matches_the_reference_module():
    import sst_amd
    g = load_golden('voxel_mixer.npz')
    net = sst_amd.BACKBONES.build(dict(type='VirtualVoxelMixer', **MIXER_CFG))
    mine, ref = net.state_dict(), _state(g)
    assert set(mine) == set(ref) and all(tuple(mine[k].shape) == tuple(ref[k].shape) for k in ref)


@pytest.mark.gpu
def test_virtual_voxel_mixer_matches_reference_golden():
    """FSDv2's backbone (sparse_unet.py:417-504), training mode, against the reference's own class run on CPU"""
    import sst_amd
    g = load_golden('voxel_mixer.npz')
    net = sst_amd.VirtualVoxelMixer(**MIXER_CFG)
    net.load_state_dict(_state(g), strict=True)
    net = net.to(DEV).train()
    x = torch.from_numpy(g['in::features']).to(DEV).requires_grad_(True)
    ind = torch.from_numpy(g['in::indices']).to(DEV)
    feats, out_ind, shape = net(x, ind, 2)
    assert torch.equal(out_ind, ind) and list(shape) == MIXER_CFG['sparse_shape']
    (feats * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()
    params = dict(net.named_parameters())
    for got, key in ((feats, 'out::features'), (x.grad, 'out::grad_features'),
                     (params['conv_out.0.weight'].grad, 'out::grad::conv_out.0.weight'),
                     (params['encoder_layers.encoder_layer2.0.0.weight'].grad,
                      'out::grad::encoder_layers.encoder_layer2.0.0.weight')):
        want = torch.from_numpy(g[key])
        err = float((got.detach().cpu() - want).abs().max())
        assert err < 1e-3 * max(1.0, float(want.abs().max())), (key, err)
