"""CPU: the host-only size queries and layout helpers added in round 4 (no kernel is launched): work list of the segmented
reduction, workspace of the thin-level sparse convolution, workspace of the encoder layer's backward call, the slab of tensors
a layer keeps, and the argument structs of the layer entry points against the header."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from sst_amd import _lib
    return _lib.load()


def test_segment_reduce_work_list_size():
    lib = _lib()
    w = lib.sst_segment_reduce_work_words
    base = w(116000, 90107, 128)
    assert base > 8 + 3 * 90107                      # header + one entry per group at least
    assert w(116000, 90107, 64) < base < w(232000, 90107, 128)
    assert w(1000, 10, 3) > 0 and w(0, 0, 1) > 0
    assert w(-1, 10, 4) < 0 and w(10, 10, 0) < 0
    # list capacities are multiples of 4 words: the partial records behind them stay 16-byte aligned
    for n, m, c in ((116000, 90107, 128), (173000, 18313, 64), (5, 3, 10)):
        chunks = n // 512 + 1
        cap_e, cap_m = -(-(m + chunks) // 4) * 4, -(-chunks // 4) * 4
        cpad = (c + 3) // 4 * 4
        assert w(n, m, c) == 8 + 3 * cap_e + 3 * cap_m + 2 * (2 * chunks) * cpad
        assert (8 + 3 * cap_e + 3 * cap_m) % 4 == 0


def test_thin_level_convolution_workspace():
    lib = _lib()
    pack = lib.sst_spconv_conv_os_f32x6_workspace_bytes(27, 256, 256)
    thin = lib.sst_spconv_conv_os_f32x6_workspace_bytes_rows(27, 256, 256, 1866)
    wide = lib.sst_spconv_conv_os_f32x6_workspace_bytes_rows(27, 64, 64, 182661)
    assert thin >= pack + 8 * 1866 * 256 * 4          # eight partial tiles per row at that size
    assert wide < lib.sst_spconv_conv_os_f32x6_workspace_bytes(27, 64, 64) + 4096      # a wide level is not split
    assert lib.sst_spconv_conv_os_f32x6_workspace_bytes_rows(27, 64, 64, -1) < 0


def test_encoder_layer_backward_workspace_and_slab():
    lib = _lib()
    from sst_amd import sst_basic_block as B
    ws = lib.sst_encoder_layer_bwd_workspace_bytes(90107, 8)
    parts = [lib.sst_add_layernorm_bwd_workspace_bytes(90107, 128), lib.sst_sra_attn_bwd_workspace_bytes(90107, 8)]
    assert ws > sum(parts) and lib.sst_encoder_layer_bwd_workspace_bytes(-1, 8) < 0
    offs, total = B._slab_offsets(90107)
    assert list(offs) == [name for name, _ in B._SLAB]
    assert all(o % 256 == 0 for o in offs.values()) and total % 256 == 0
    names = [n for n, _ in B._SLAB]
    for (a, cols), b in zip(B._SLAB[:-1], names[1:]):
        assert offs[b] - offs[a] >= 90107 * cols * 4


@pytest.mark.parametrize('struct,cname', [('EncoderLayerFwdArgs', 'sst_encoder_layer_fwd_args'),
                                          ('EncoderLayerBwdArgs', 'sst_encoder_layer_bwd_args'),
                                          ('EncoderLayerFwdBF16Args', 'sst_encoder_layer_fwd_bf16_args'),
                                          ('EncoderLayerBwdBF16Args', 'sst_encoder_layer_bwd_bf16_args')])
def test_layer_argument_structs_match_the_header(struct, cname):
    """field names and order of the ctypes structures == the typedefs of include/sst_amd.h"""
    from sst_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'sst_amd.h')).read()
    body = re.search(r'typedef struct ' + cname + r' \{(.*?)\} ' + cname + ';', text, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)      # field comments
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r'^(const\s+)?(int64_t|int32_t|float|void)\b', '', decl)
        fields += [f.strip().lstrip('*').strip() for f in decl.split(',')]
    assert fields == [f[0] for f in getattr(_lib, struct)._fields_]
    s = getattr(_lib, struct)
    assert ctypes.sizeof(s) == 16 + 16 + 8 + 8 * (len(fields) - 8)


def test_fused_vfe_declines_what_it_is_not_built_for():
    import sst_amd
    from sst_amd.vfe_fused import fused_vfe2_ok
    vfe = sst_amd.build_voxel_encoder(dict(
        type='DynamicVFE', in_channels=3, feat_channels=[64, 128], with_cluster_center=True, with_voxel_center=True,
        voxel_size=(0.32, 0.32, 6), point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)))

    class Plan(object):
        raw_max = group_sum = staticmethod(lambda *a, **k: None)
        num_voxels = 5
    x = torch.zeros(10, 9)
    assert not fused_vfe2_ok(vfe, x, Plan())                      # a CPU tensor
    three = sst_amd.build_voxel_encoder(dict(
        type='DynamicVFE', in_channels=3, feat_channels=[64, 64, 128], with_cluster_center=True, with_voxel_center=True,
        voxel_size=(0.32, 0.32, 6), point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4]))
    assert len(three.vfe_layers) == 3 and not fused_vfe2_ok(three, x, Plan())


def test_precision_switch_is_scoped_to_the_module():
    """two backbones of one process with different set_precision(): each runs its encoder stack in ITS mode, whichever call
    came last (the switch used to be process-global); set_precision() touches nothing process-wide, and a
    backbone nobody called set_precision() on multiplies from the exact split ('f32x6', the default since round 5)"""
    import sst_amd
    from sst_amd import dense
    cfg = dict(type='SSTv2', d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2, output_shape=[468, 468],
               debug=False, num_attached_conv=0, to_bev=False)
    a, b, c = sst_amd.build_backbone(dict(cfg)), sst_amd.build_backbone(dict(cfg)), sst_amd.build_backbone(dict(cfg))
    assert dense.DEFAULT_MATMUL_MODE == 'f32x6' and dense.matmul_mode() == 'f32x6'
    a.set_precision('f32x3')
    b.set_precision('fp32')                      # the opt-out onto the fp32 matrix pipe: this module only
    assert dense.matmul_mode() == 'f32x6' and a.matmul == 'f32x3' and b.matmul == 'f32' and c.matmul == 'f32x6'
    assert c.precision == 'fp32' and b.set_precision('bf16').precision == 'bf16' and b.matmul == 'f32'
    b.set_precision('fp32')
    seen = {}
    for name, model in (('a', a), ('b', b), ('c', c)):
        model._window_inputs = lambda info: (None, None, None)
        model.run_blocks = lambda feats, pos, plans, masks=None, pos_lookup=None, _n=name: seen.setdefault(_n, dense.matmul_mode()) and feats
        model({'voxel_coors': torch.zeros((4, 4), dtype=torch.int64), 'voxel_feats': torch.zeros(4, 128)})
    assert seen == {'a': 'f32x3', 'b': 'f32', 'c': 'f32x6'} and dense.matmul_mode() == 'f32x6'
    with pytest.raises(ValueError):
        with dense.matmul_mode_scope('tf32'):
            pass
