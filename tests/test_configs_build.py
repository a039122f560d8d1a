"""The drop-in contract at its entry: the reference's shipped config files, read from their text, construct through
sst_amd's registries UNMODIFIED.  Build container only (the config text lives
under /root/reference); construction is pure host work, no kernel runs.

  configs/fsdv2/*.py          SingleStageFSDV2 / FSDV2 with multiscale_cfg (all five) and as_rpn (Waymo): BASELINE configs[4]
  configs/fsd/*.py            FSD with the sparse-convolution or the SST segmentor: BASELINE configs[3]
  configs/sst_refactor/*.py,  voxel_layer / voxel_encoder / middle_encoder / backbone of DynamicVoxelNet / DynamicCenterPoint:
  configs/sst/*.py            BASELINE configs[1..2]
and bench_workloads.FSDV2_CFG (what `bench.py --workload fsdv2` runs) is compared number by number with fsdv2_nusc_1x.py."""
import copy
import glob
import os

import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'configs')), reason='needs the reference tree')


def _merge(base, new):
    """mmcv.Config._merge_a_into_b: dictionaries merge recursively, `_delete_=True` replaces"""
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy({kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v)
    return out


def load_config(path):
    """a config file as mmcv.Config.fromfile reads it: exec the text, resolve `_base_` files first"""
    ns = {}
    exec(compile(open(path).read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop('_base_', [])
    merged = {}
    for b in ([bases] if isinstance(bases, str) else bases):
        merged = _merge(merged, load_config(os.path.normpath(os.path.join(os.path.dirname(path), b))))
    return _merge(merged, cfg)


def _configs(sub):
    return sorted(p for p in glob.glob(os.path.join(REF, 'configs', sub, '*.py')) if 'do_not_use' not in p)


@pytest.mark.parametrize('path', _configs('fsdv2'), ids=os.path.basename)
def test_fsdv2_configs_construct(path):
    import sst_amd
    model = load_config(path)['model']
    det = sst_amd.build_detector(model)
    assert type(det).__name__ == model['type'] and isinstance(det, sst_amd.VirtualVoxelExtractor)
    keys = set(det.state_dict())
    ms = model['multiscale_cfg']
    assert len(det.ms_projectors) == len(ms['projector_hiddens']) == 3
    for i, proj in enumerate(ms['projector_hiddens']):
        assert det.state_dict()[f'ms_projectors.{i}.0.0.weight'].shape == (proj[1], proj[0])
    assert det.segmentor.use_multiscale_features and det.segmentor.backbone.return_multiscale_features
    assert det.as_rpn == bool(model['bbox_head'].get('as_rpn', False))
    assert ('recover_proj.0.0.weight' in keys) == det.as_rpn
    for prefix in ('segmentor.voxel_encoder.', 'segmentor.backbone.conv_input.', 'segmentor.segmentation_head.conv_seg.',
                   'segmentor.segmentation_head.voting.', 'segmentor.segmentation_head.pre_seg_conv.', 'virtual_proj.',
                   'ori_proj.', 'voxel_encoder.vfe_layers.', 'backbone.conv_out.'):
        assert any(k.startswith(prefix) for k in keys), prefix
    assert 'bbox_head' in det.unbuilt and all(not k.startswith('bbox_head') for k in keys)       # heads: out of scope, not built
    if model['type'] == 'FSDV2':
        assert isinstance(det.roi_extractor, sst_amd.DynamicPointROIExtractor)
    # the target grid of the fusion is the mixer's grid, and every fused level maps into it with integer strides
    assert list(ms['target_sparse_shape']) == list(model['backbone']['sparse_shape'])


@pytest.mark.parametrize('path', _configs('fsd'), ids=os.path.basename)
def test_fsd_configs_construct(path):
    import sst_amd
    model = load_config(path)['model']
    det = sst_amd.build_detector(model)
    assert type(det).__name__ == model['type']
    keys = set(det.state_dict())
    if model['type'] == 'VoteSegmentor':        # the segmentor-only pre-training config
        assert any(k.startswith('backbone.') for k in keys) and any(k.startswith('segmentation_head.voting.') for k in keys)
        return
    assert isinstance(det.backbone, sst_amd.SIR) and isinstance(det.cluster_assigner, sst_amd.ClusterAssigner)
    assert det.cluster_assigner.num_classes == model['bbox_head']['num_classes']
    assert any(k.startswith('backbone.block_list.2.vfe_layers.1.') for k in keys)
    assert any(k.startswith('segmentor.backbone.') for k in keys)
    if 'gpu_clustering' in model['cluster_assigner']:
        assert tuple(det.cluster_assigner.gpu_clustering) == tuple(model['cluster_assigner']['gpu_clustering'])


@pytest.mark.parametrize('path', _configs('sst_refactor') + _configs('sst'), ids=os.path.basename)
def test_sst_configs_construct(path):
    """the four hot-path sub-configs of DynamicVoxelNet / DynamicCenterPoint (detectors/dynamic_voxelnet.py:38-71)"""
    import sst_amd
    model = load_config(path)['model']
    layer = sst_amd.Voxelization(**model['voxel_layer'])
    vfe = sst_amd.build_voxel_encoder(model['voxel_encoder'])
    mid = sst_amd.build_middle_encoder(model['middle_encoder'])
    bb = sst_amd.build_backbone(model['backbone'])
    assert isinstance(vfe, sst_amd.DynamicVFE) and layer is not None and mid is not None
    d_model = model['backbone']['d_model'][0]
    sd = bb.state_dict()
    assert sd['block_list.0.encoder_list.0.win_attn.self_attn.in_proj_weight'].shape == (3 * d_model, d_model)
    assert len(bb.block_list) == model['backbone']['num_blocks']


def test_bench_fsdv2_config_is_the_shipped_nuscenes_config():
    """`bench.py --workload fsdv2` (bench_workloads.FSDV2_CFG) against configs/fsdv2/fsdv2_nusc_1x.py, number by number"""
    import bench_workloads as BW
    cfg = load_config(os.path.join(REF, 'configs/fsdv2/fsdv2_nusc_1x.py'))
    model, seg, mine = cfg['model'], cfg['model']['segmentor'], BW.FSDV2_CFG
    assert tuple(mine['seg_voxel']) == tuple(seg['voxel_layer']['voxel_size'])
    assert list(mine['pc_range']) == list(seg['voxel_layer']['point_cloud_range'])
    assert tuple(mine['virtual_voxel']) == tuple(model['voxel_encoder']['voxel_size'])
    for k, v in mine['vfe'].items():
        assert seg['voxel_encoder'][k] == v, k
    for k, v in mine['unet'].items():
        assert _plain(seg['backbone'][k]) == _plain(v), k
    assert seg['backbone']['return_multiscale_features'] is True
    for k, v in mine['mixer'].items():
        assert _plain(model['backbone'][k]) == _plain(v), k
    assert mine['virtual_vfe']['feat_channels'] == model['voxel_encoder']['feat_channels']
    assert mine['proj_hidden'] == model['virtual_point_projector']['hidden_dims'] == model['virtual_point_projector']['ori_hidden_dims']
    for k in ('multiscale_levels', 'projector_hiddens', 'fusion_mode', 'target_sparse_shape', 'norm_cfg'):
        assert _plain(mine['multiscale'][k]) == _plain(model['multiscale_cfg'][k]), k
    assert mine['as_rpn'] == bool(model['bbox_head'].get('as_rpn', False))
    assert mine['n_logits'] == seg['segmentation_head']['num_classes'] + 1       # softmax head: + background


def _plain(v):
    if isinstance(v, (list, tuple)):
        return [_plain(e) for e in v]
    if isinstance(v, dict):
        return {k: _plain(e) for k, e in v.items()}
    return v
