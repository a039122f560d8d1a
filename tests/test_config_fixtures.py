"""The resolved `model` dictionaries under tests/golden/configs/ (what `bench.py`'s config_as_is leg and `--workload sst_center`
build from on the GPU box, where /root/reference does not exist): they equal what the reference's config files say (build
container only), and they construct through `sst_amd.build_detector` UNMODIFIED - no extra keyword, no call after construction -
into the detector whose `extract_feat` is the hot path (dynamic_voxelnet.py:38-47).  Host work only."""
import ast
import glob
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, 'golden', 'configs', '*.model.py')))


def load_fixture(path):
    return ast.literal_eval(open(path).read())


def test_there_are_fixtures():
    assert len(FIXTURES) == 3


@pytest.mark.skipif(not os.path.isdir('/root/reference/configs'), reason='needs the reference tree')
@pytest.mark.parametrize('path', FIXTURES, ids=os.path.basename)
def test_fixture_is_the_shipped_config(path):
    from test_configs_build import REF, load_config, _plain
    name = os.path.basename(path)[:-len('.model.py')]
    want = load_config(os.path.join(REF, 'configs', 'sst_refactor', name + '.py'))['model']
    got = load_fixture(path)
    assert got == want                     # == on dicts / tuples / lists is type-strict for tuple vs list
    assert repr(_plain(got)) == repr(_plain(want))


@pytest.mark.parametrize('path', FIXTURES, ids=os.path.basename)
def test_fixture_constructs_as_shipped(path):
    import sst_amd
    from sst_amd import dense
    model = load_fixture(path)
    det = sst_amd.build_detector(model)
    assert type(det).__name__ == model['type'] and isinstance(det, sst_amd.DynamicVoxelNet)
    assert isinstance(det.voxel_encoder, sst_amd.DynamicVFE) and isinstance(det.middle_encoder, sst_amd.SSTInputLayerV2)
    assert isinstance(det.backbone, sst_amd.SSTv2) and len(det.backbone.block_list) == model['backbone']['num_blocks']
    assert set(det.unbuilt) == {'neck', 'bbox_head'}            # dense neck + box head: out of scope, kept as configs
    keys = set(det.state_dict())
    assert 'backbone.block_list.0.encoder_list.0.win_attn.self_attn.in_proj_weight' in keys
    assert 'voxel_encoder.vfe_layers.0.linear.weight' in keys
    assert ('backbone.block_list.0.encoder_list.0.win_attn.self_attn.tau' in keys) == \
        bool(model['backbone'].get('layer_cfg', {}).get('cosine', False))
    # the defaults ARE the fast path: exact-split products, window-major voxel order, reference-style entries on demand, fused plan
    assert det.backbone.matmul == dense.DEFAULT_MATMUL_MODE == 'f32x6' and det.backbone.precision == 'fp32'
    me = det.middle_encoder
    assert me.reference_outputs is True and me.window_major is True and me.shuffle_voxels is True and me.debug is True
    assert det.fused_index is True


def test_voxel_info_defers_the_reference_entries():
    from sst_amd.sst_input_layer import VoxelInfo
    formed = []
    info = VoxelInfo(voxel_feats=1)
    info.defer(['flat2win_inds_shift0', 'key_mask_shift0'], lambda d: (formed.append(1), dict.update(d, flat2win_inds_shift0='a',
                                                                                                      key_mask_shift0='b')))
    assert 'key_mask_shift0' in info and info.get('nope') is None and info.peek('key_mask_shift0') is None and not formed
    assert len(info) == 3
    copy = info.shallow()
    assert info['flat2win_inds_shift0'] == 'a' and info['key_mask_shift0'] == 'b' and formed == [1]
    assert copy.peek('key_mask_shift0') is None and dict(copy)['key_mask_shift0'] == 'b' and formed == [1, 1]
    with pytest.raises(KeyError):
        info['missing']
    assert sorted(info.keys()) == ['flat2win_inds_shift0', 'key_mask_shift0', 'voxel_feats']


class _ReferenceLikeDetector(torch.nn.Module):
    """what the reference's DynamicVoxelNet instance looks like after the registry swap of INTEGRATION.md section A: the four
    sub-modules are this library's, the class itself (voxelize loop, neck, heads) is the reference's"""

    def __init__(self, cfg):
        super().__init__()
        import sst_amd
        self.voxel_layer = sst_amd.Voxelization(**cfg['voxel_layer'])
        self.voxel_encoder = sst_amd.build_voxel_encoder(cfg['voxel_encoder'])
        self.middle_encoder = sst_amd.build_middle_encoder(cfg['middle_encoder'])
        self.backbone = sst_amd.build_backbone(cfg['backbone'])
        self.neck, self.with_neck = torch.nn.Identity(), True

    @torch.no_grad()
    def voxelize(self, points):            # dynamic_voxelnet.py:49-71, verbatim semantics: per-sample loop, pad, concatenate
        coors = [torch.nn.functional.pad(self.voxel_layer(p), (1, 0), mode='constant', value=i) for i, p in enumerate(points)]
        return torch.cat(points, dim=0), torch.cat(coors, dim=0)


def test_install_fused_extract_feat_on_a_reference_like_class():
    import sst_amd
    cfg = load_fixture(FIXTURES[0])
    cls = sst_amd.install_fused_extract_feat(type('Det', (_ReferenceLikeDetector,), {}))
    det = cls(cfg)
    assert 'fused index plan' in cls.extract_feat.__doc__
    planner = sst_amd.DynamicVoxelNet._frame_planner(det, 1)
    assert planner is not None and det.__dict__['_planner'] is planner and planner.supported(2) and not planner.supported(65)
    det.fused_index = False
    assert sst_amd.DynamicVoxelNet._frame_planner(det, 1) is None
    assert sst_amd.DynamicVoxelNet.prepare(det, [torch.zeros(5, 3)]) is None         # CPU clouds: the piecewise path
