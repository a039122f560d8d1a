"""TEST INFRASTRUCTURE ONLY — loads the reference's own Python (unmodified, by file path from
/root/reference) under small stub modules, so that the restatements in oracle/ can be checked against
it and golden vectors can be generated (tests/golden/make_golden.py).

/root/reference exists only in the build container, never on the GPU box: nothing under tests/ marked
``gpu``, bench.py or __graft_entry__.smoke() imports this module.

Stubbed third-party modules (absent here):
  mmcv.runner.{auto_fp16, force_fp32}  -> identity decorators
  mmcv.cnn.{build_norm_layer, build_conv_layer, NORM_LAYERS}
  ipdb.set_trace                        -> no-op
  torch_scatter.{scatter_max, scatter}  -> restated with Tensor.scatter_reduce_ (parity unpinned: the
                                          reference has no test for them; contract = segmented reduce)
  ingroup_indices.forward               -> stable-sort rank (TorchEx is un-vendored; the reference's own
                                          fallback get_inner_win_inds_deprecated defines the contract:
                                          a bijection onto 0..cnt-1 per group, order unspecified)
  mmdet.models.BACKBONES, mmdet3d.ops.spconv / make_sparse_convmodule -> placeholders (load_reference);
  load_reference_spconv() replaces the spconv placeholder by the reference's own vendored spconv Python package
  (structure / modules / ops / functional / conv .py, unmodified) on top of a stub `sparse_conv_ext` whose rulebook
  comes from the reference's CPU templates compiled into oracle/_ref and whose indice_conv is oracle/spconv_oracle's
  arithmetic in torch; mmdet 2.14's resnet.BasicBlock constructor is restated for ops/sparse_block.py
  DynamicScatter (GPU-only in the reference) -> oracle.voxel_oracle restatement
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('SST_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'mmdet3d'))


class _Registry(object):
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, **kw):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop('type')](**cfg)


def _identity_decorator(*dargs, **dkwargs):
    def deco(fn):
        return fn
    return deco


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name):
    m = _mod(name)
    m.__path__ = []
    return m


def _load(modname, relpath):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def stable_ingroup_rank(group_inds):
    """rank of each element among equal ids, in ascending index order."""
    order = torch.argsort(group_inds, stable=True)
    sorted_g = group_inds[order]
    n = group_inds.numel()
    idx = torch.arange(n, device=group_inds.device)
    is_head = torch.ones(n, dtype=torch.bool, device=group_inds.device)
    is_head[1:] = sorted_g[1:] != sorted_g[:-1]
    head_pos = torch.where(is_head, idx, torch.zeros_like(idx))
    head_pos = torch.cummax(head_pos, 0)[0]
    rank_sorted = idx - head_pos
    out = torch.empty_like(group_inds)
    out[order] = rank_sorted
    return out


def _scatter_max(src, index, dim=0):
    assert dim == 0
    m = int(index.max().item()) + 1 if index.numel() else 0
    out = torch.full((m,) + tuple(src.shape[1:]), float('-inf'), dtype=src.dtype, device=src.device)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = out.scatter_reduce(0, idx, src, reduce='amax', include_self=True)
    return out, None


def _scatter(src, index, dim=0, reduce='sum'):
    assert dim == 0
    m = int(index.max().item()) + 1 if index.numel() else 0
    out = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    red = {'sum': 'sum', 'mean': 'mean', 'add': 'sum'}[reduce]
    return out.scatter_reduce(0, idx, src, reduce=red, include_self=False)


_LOADED = None


def load_reference():
    """Returns a namespace with the reference's hot-path modules (executed from their own files)."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError('reference tree not found at ' + REF_ROOT)
    from oracle import voxel_oracle

    # ---- third-party stubs ----
    _mod('ipdb', set_trace=lambda *a, **k: None)
    _pkg('mmcv')
    _mod('mmcv.runner', auto_fp16=_identity_decorator, force_fp32=_identity_decorator)
    norm_layers = _Registry('norm')

    def build_norm_layer(cfg, num_features, postfix=''):
        cfg = dict(cfg)
        t = cfg.pop('type')
        cfg.pop('requires_grad', None)
        table = {'BN1d': nn.BatchNorm1d, 'BN2d': nn.BatchNorm2d, 'BN': nn.BatchNorm2d, 'LN': nn.LayerNorm}
        table.update(norm_layers.module_dict)
        cfg.setdefault('eps', 1e-5)
        return ('ln' if t == 'LN' else 'bn') + str(postfix), table[t](num_features, **cfg)

    conv_layers = _Registry('conv layer')

    def build_conv_layer(cfg, *args, **kwargs):
        cfg = dict(cfg or dict(type='Conv2d'))
        t = cfg.pop('type')
        table = {'Conv2d': nn.Conv2d, 'Conv1d': nn.Conv1d}
        table.update(conv_layers.module_dict)
        return table[t](*args, **kwargs, **cfg)

    _mod('mmcv.cnn', build_norm_layer=build_norm_layer, build_conv_layer=build_conv_layer, NORM_LAYERS=norm_layers,
         CONV_LAYERS=conv_layers)
    _mod('torch_scatter', scatter_max=_scatter_max, scatter=_scatter)

    def _ingroup_forward(group_inds, out_inds):
        out_inds.copy_(stable_ingroup_rank(group_inds))

    _mod('ingroup_indices', forward=_ingroup_forward)
    _pkg('mmdet')
    backbones = _Registry('backbone')
    _mod('mmdet.models', BACKBONES=backbones)

    # ---- mmdet3d package skeleton ----
    _pkg('mmdet3d')
    ops = _pkg('mmdet3d.ops')
    ops.spconv = types.SimpleNamespace()
    ops.make_sparse_convmodule = None
    _pkg('mmdet3d.ops.sst')
    models = _pkg('mmdet3d.models')
    models_reg = _Registry('models')
    builder = _mod('mmdet3d.models.builder', MODELS=models_reg, VOXEL_ENCODERS=models_reg,
                   MIDDLE_ENCODERS=models_reg, BACKBONES=backbones,
                   build_voxel_encoder=models_reg.build, build_middle_encoder=models_reg.build,
                   build_backbone=backbones.build, build_fusion_layer=None)
    models.builder = builder

    # reference files, executed from where they lie
    norm = _load('mmdet3d.ops.norm', 'mmdet3d/ops/norm.py')
    sst_ops = _load('mmdet3d.ops.sst.sst_ops', 'mmdet3d/ops/sst/sst_ops.py')
    for name in ('flat2window', 'window2flat', 'get_flat2win_inds', 'get_inner_win_inds', 'make_continuous_inds',
                 'flat2window_v2', 'window2flat_v2', 'get_flat2win_inds_v2', 'get_window_coors', 'scatter_v2',
                 'build_mlp', 'get_activation', 'get_activation_layer'):
        setattr(ops, name, getattr(sst_ops, name))
    ops.DynamicScatter = voxel_oracle.DynamicScatterOracle  # GPU-only in the reference (voxelization.h:106)
    ops.NaiveSyncBatchNorm1d = norm.NaiveSyncBatchNorm1d

    _pkg('mmdet3d.models.middle_encoders')
    _pkg('mmdet3d.models.sst')
    _pkg('mmdet3d.models.backbones')
    _pkg('mmdet3d.models.voxel_encoders')
    ns = types.SimpleNamespace()
    ns.sst_ops = sst_ops
    ns.norm = norm
    ns.input_layer_v2 = _load('mmdet3d.models.middle_encoders.sst_input_layer_v2',
                              'mmdet3d/models/middle_encoders/sst_input_layer_v2.py')
    ns.input_layer_v1 = _load('mmdet3d.models.middle_encoders.sst_input_layer',
                              'mmdet3d/models/middle_encoders/sst_input_layer.py')
    ns.block_v1 = _load('mmdet3d.models.sst.sst_basic_block', 'mmdet3d/models/sst/sst_basic_block.py')
    ns.sst_v1 = _load('mmdet3d.models.backbones.sst_v1', 'mmdet3d/models/backbones/sst_v1.py')
    ns.cosine_msa = _load('mmdet3d.models.sst.cosine_msa', 'mmdet3d/models/sst/cosine_msa.py')
    ns.block_v2 = _load('mmdet3d.models.sst.sst_basic_block_v2', 'mmdet3d/models/sst/sst_basic_block_v2.py')
    ns.sst_v2 = _load('mmdet3d.models.backbones.sst_v2', 'mmdet3d/models/backbones/sst_v2.py')
    ns.vfe_utils = _load('mmdet3d.models.voxel_encoders.utils', 'mmdet3d/models/voxel_encoders/utils.py')
    ns.voxel_encoder = _load('mmdet3d.models.voxel_encoders.voxel_encoder',
                             'mmdet3d/models/voxel_encoders/voxel_encoder.py')
    ns.sir = _load('mmdet3d.models.backbones.sir', 'mmdet3d/models/backbones/sir.py')
    ns.builder = builder
    ns.backbones = backbones
    _LOADED = ns
    return ns


_SPCONV = None


def load_reference_spconv():
    """The reference's vendored spconv Python package + ops/sparse_block.py + middle_encoders/sparse_unet.py, executed
    unmodified on CPU.  Native side (`sparse_conv_ext`, CUDA / extension code in the reference): get_indice_pairs_3d
    = the reference's CPU rulebook templates compiled by oracle/build_ref.build_spconv_rulebook (subm forces stride 1
    and padding ksize // 2 as spconv_ops.h:74-77 does); indice_conv(_backward)_fp32 = the per-offset gather / mm /
    scatter-add of spconv_ops.h:256-446 written with torch CPU ops."""
    global _SPCONV
    if _SPCONV is not None:
        return _SPCONV
    ns = load_reference()
    from oracle import build_ref
    build_ref.build_spconv_rulebook()
    rule = build_ref.load_spconv_rulebook()
    assert rule is not None

    def get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation,
                            out_padding, subm, transpose):
        if subm:
            stride, padding = [1] * 3, [k // 2 for k in ksize]
        outids, pairs, num = rule.get_indice_pairs_3d(indices.int().contiguous(), int(batch_size), list(out_shape),
                                                      list(ksize), list(stride), list(padding), list(dilation),
                                                      bool(subm), bool(transpose))
        return [outids, pairs, num]

    def indice_conv_fp32(features, filters, indice_pairs, indice_pair_num, num_act_out, inverse, subm):
        w = filters.reshape(-1, filters.shape[-2], filters.shape[-1])
        out = torch.zeros(num_act_out, w.shape[2], dtype=features.dtype)
        for k in range(w.shape[0]):
            c = int(indice_pair_num[k])
            if c:
                src, dst = indice_pairs[k, int(inverse), :c].long(), indice_pairs[k, 1 - int(inverse), :c].long()
                out.index_add_(0, dst, features[src] @ w[k])
        return out

    def indice_conv_backward_fp32(features, filters, out_bp, indice_pairs, indice_pair_num, inverse, subm):
        w = filters.reshape(-1, filters.shape[-2], filters.shape[-1])
        dx = torch.zeros_like(features)
        dw = torch.zeros_like(w)
        for k in range(w.shape[0]):
            c = int(indice_pair_num[k])
            if c:
                src, dst = indice_pairs[k, int(inverse), :c].long(), indice_pairs[k, 1 - int(inverse), :c].long()
                dw[k] = features[src].t() @ out_bp[dst]
                dx.index_add_(0, src, out_bp[dst] @ w[k].t())
        return [dx, dw.reshape(filters.shape)]

    pkg = _pkg('mmdet3d.ops.spconv')
    pkg.IS_SPCONV2_AVAILABLE = False
    sys.modules['mmdet3d.ops'].spconv = pkg
    _mod('mmdet3d.ops.spconv.sparse_conv_ext', get_indice_pairs_3d=get_indice_pairs_3d,
         indice_conv_fp32=indice_conv_fp32, indice_conv_backward_fp32=indice_conv_backward_fp32)
    pkg.sparse_conv_ext = sys.modules['mmdet3d.ops.spconv.sparse_conv_ext']
    out = types.SimpleNamespace()
    for name in ('structure', 'modules', 'ops', 'functional', 'conv'):
        m = _load('mmdet3d.ops.spconv.' + name, f'mmdet3d/ops/spconv/{name}.py')
        setattr(pkg, name, m)
        setattr(out, name, m)

    class SparseConvTensor(out.structure.SparseConvTensor):
        """+ replace_feature of spconv 2.x, which sparse_unet.py calls (the vendored 1.x class lacks it)"""

        def replace_feature(self, new_features):
            t = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid)
            t.indice_dict = self.indice_dict
            return t

    out.modules.SparseConvTensor = SparseConvTensor   # SparseSequential's isinstance checks
    out.conv.SparseConvTensor = SparseConvTensor      # tensors created by the convolutions
    _mod('mmcv.ops', SparseModule=out.modules.SparseModule, SparseSequential=out.modules.SparseSequential,
         SparseConvTensor=SparseConvTensor)
    sys.modules['mmcv.runner'].BaseModule = type('BaseModule', (nn.Module,), {
        '__init__': lambda self, init_cfg=None: nn.Module.__init__(self)})

    # mmdet 2.14.0 (docs/overall_instructions.md:38) resnet.BasicBlock, constructor restated; Bottleneck unused here
    build_norm_layer = sys.modules['mmcv.cnn'].build_norm_layer
    build_conv_layer = sys.modules['mmcv.cnn'].build_conv_layer

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                     conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, plugins=None, init_cfg=None):
            nn.Module.__init__(self)
            self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
            self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
            self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation,
                                          dilation=dilation, bias=False)
            self.add_module(self.norm1_name, norm1)
            self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
            self.add_module(self.norm2_name, norm2)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample
            self.stride = stride
            self.dilation = dilation
            self.with_cp = with_cp

        @property
        def norm1(self):
            return getattr(self, self.norm1_name)

        @property
        def norm2(self):
            return getattr(self, self.norm2_name)

    _pkg('mmdet.models.backbones')
    _mod('mmdet.models.backbones.resnet', BasicBlock=BasicBlock, Bottleneck=BasicBlock)
    out.sparse_block = _load('mmdet3d.ops.sparse_block', 'mmdet3d/ops/sparse_block.py')
    ops_pkg = sys.modules['mmdet3d.ops']
    ops_pkg.SparseBasicBlock = out.sparse_block.SparseBasicBlock
    ops_pkg.make_sparse_convmodule = out.sparse_block.make_sparse_convmodule
    out.sparse_unet = _load('mmdet3d.models.middle_encoders.sparse_unet', 'mmdet3d/models/middle_encoders/sparse_unet.py')
    out.SparseConvTensor = SparseConvTensor
    out.base = ns
    _SPCONV = out
    return out


def load_reference_function(relpath, name, extra_globals=None):
    """One module-level function of a reference file, executed from the file's own source text (for files whose
    imports cannot be satisfied here, e.g. detectors/single_stage_fsd.py needs mmdet / mmseg).  Nothing is copied
    into the repository."""
    import ast
    import textwrap
    path = os.path.join(REF_ROOT, relpath)
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            code = textwrap.dedent(ast.get_source_segment(src, node))
            glb = {'torch': torch}
            glb.update(extra_globals or {})
            exec(compile(code, path + ':' + name, 'exec'), glb)
            return glb[name]
    raise KeyError(name)



def load_reference_method(relpath, cls_name, name, extra_globals=None):
    """One method of a class of a reference file, as a plain function taking ``self`` - executed from the file's own
    source text (the detector modules import mmdet / mmseg and cannot be imported here).  Nothing is copied."""
    import ast
    import textwrap
    path = os.path.join(REF_ROOT, relpath)
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == name:
                    lines = src.splitlines()[item.lineno - 1:item.end_lineno]     # without the decorators
                    code = textwrap.dedent('\n'.join(lines))
                    glb = {'torch': torch}
                    glb.update(extra_globals or {})
                    exec(compile(code, path + ':' + cls_name + '.' + name, 'exec'), glb)
                    return glb[name]
    raise KeyError(cls_name + '.' + name)


def load_reference_class(relpath, cls_name, extra_globals=None):
    """One class of a reference file, executed from the file's own source text (for files that cannot be imported here,
    e.g. detectors/single_stage_fsd.py needs mmdet / mmseg).  Decorators are dropped (registry registration); nothing is
    copied into the repository."""
    import ast
    path = os.path.join(REF_ROOT, relpath)
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            lines = src.splitlines()[node.lineno - 1:node.end_lineno]      # from the `class` line on: no decorators
            glb = {'torch': torch, 'nn': nn}
            glb.update(extra_globals or {})
            exec(compile('\n'.join(lines), path + ':' + cls_name, 'exec'), glb)
            return glb[cls_name]
    raise KeyError(cls_name)
