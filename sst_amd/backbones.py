"""SST and SIR backbones on the window-CSR kernels.

Drop-in for the reference's registry entries 'SSTv2' (mmdet3d/models/backbones/sst_v2.py:16-196), 'SSTv1'
(backbones/sst_v1.py:17-270) and 'SIR' (backbones/sir.py:15-87): constructor keywords, forward signatures and
``state_dict`` keys (``block_list.{i}.encoder_list.{0,1}.*``, ``conv_layer.{j}.{0,1}.*``, ``linear0.*``) are the
reference's, so its configs and checkpoints load unchanged.  The bodies are organised around this package's plan
objects instead of the reference's per-level dictionaries:

* both SST generations share one stem (`_WindowTransformer`): a stack of BasicShiftBlockV2 driven by two
  ``kernels.WindowPlan`` (window CSR of the regular / shifted partition) and two flat [M, C] positional tensors -
  whatever the input layer handed over (plans of this package, or the reference's flat2win dictionaries, which are
  converted once per forward);
* the dense BEV canvas is written by one kernel pass (csrc/scatter.hip, ``sst_recover_bev_f32``: every cell once,
  channels-last) and handed on as the channels-last view of the reference's [B, C, ny, nx] tensor;
* SIR groups the points of a cluster ONCE (``UniquePlan``) and every layer's pooling reuses it.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from .norm import build_conv_layer, build_norm_layer
from .registry import BACKBONES, build_voxel_encoder
from .sst_basic_block import BasicShiftBlockV2, plan_from_reference_dicts
from .sst_ops import unique_with_plan


class _RecoverBEV(torch.autograd.Function):
    """[M, C] voxel rows -> [B, ny, nx, C] canvas (csrc/scatter.hip); backward gathers the rows back."""

    @staticmethod
    def forward(ctx, feats, coors, batch, ny, nx):
        lib = _lib.load()
        feats = feats.contiguous()
        coors = coors.contiguous()
        m, c = feats.shape
        dev = feats.device
        canvas = torch.empty((batch, ny, nx, c), dtype=torch.float32, device=dev)
        cell_map = torch.empty(batch * ny * nx, dtype=torch.int32, device=dev)
        cell_of_voxel = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
        rc = lib.sst_recover_bev_f32(_lib.ptr(feats), feats.stride(0), _lib.ptr(coors),
                                     int(coors.dtype == torch.int64), coors.stride(0), m, batch, ny, nx, c,
                                     _lib.ptr(cell_map), _lib.ptr(cell_of_voxel), _lib.ptr(canvas), _lib.stream_ptr())
        _lib.check(rc, 'sst_recover_bev_f32')
        ctx.save_for_backward(cell_of_voxel)
        ctx.shape = (m, c)
        return canvas

    @staticmethod
    def backward(ctx, grad_canvas):
        (cell_of_voxel,) = ctx.saved_tensors
        m, c = ctx.shape
        grad_canvas = grad_canvas.contiguous()   # [B, ny, nx, C]
        out = torch.empty((m, c), dtype=torch.float32, device=grad_canvas.device)
        rc = _lib.load().sst_recover_bev_bwd_f32(_lib.ptr(grad_canvas), _lib.ptr(cell_of_voxel), m, c, _lib.ptr(out),
                                                 _lib.stream_ptr())
        _lib.check(rc, 'sst_recover_bev_bwd_f32')
        return out, None, None, None, None


def recover_bev(voxel_feat, coors, batch_size, output_shape):
    """Dense [B, C, ny, nx] canvas of the voxel features (sst_v2.py:161-197), returned as the channels-last view of that
    logical shape.  Shapes the kernel is not built for (C % 4 != 0, CPU tensors) take an index assignment in torch."""
    ny, nx = output_shape
    c = voxel_feat.shape[-1]
    if voxel_feat.is_cuda and voxel_feat.dtype == torch.float32 and c % 4 == 0 and coors.dtype in (torch.int32, torch.int64):
        return _RecoverBEV.apply(voxel_feat, coors, int(batch_size), int(ny), int(nx)).permute(0, 3, 1, 2)
    flat = (coors[:, 0] * ny + coors[:, 2]) * nx + coors[:, 3]
    canvas = voxel_feat.new_zeros((batch_size * ny * nx, c)).index_put((flat.long(),), voxel_feat)
    return canvas.view(batch_size, ny, nx, c).permute(0, 3, 1, 2)


def _sparse_first_conv_ok(conv, voxel_feat):
    """the first attached convolution can run on the voxel rows themselves: a plain 2-D convolution that keeps the canvas
    size, fp32 CUDA rows of a width the sparse-convolution kernels take"""
    if os.environ.get('SST_BEV_SPARSE_FIRST_CONV', '1') == '0' or not isinstance(conv, nn.Conv2d):
        return False
    k, d, p = conv.kernel_size, conv.dilation, conv.padding
    return (conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros'
            and not isinstance(p, str) and all(2 * p[i] == d[i] * (k[i] - 1) for i in range(2)) and k[0] * k[1] <= 32
            and voxel_feat.is_cuda and voxel_feat.dtype == torch.float32 and voxel_feat.size(1) % 4 == 0
            and voxel_feat.size(0) > 0)


def sparse_first_conv(voxel_feat, coors, batch_size, output_shape, conv):
    """``conv(recover_bev(voxel_feat))`` WITHOUT the dense input canvas (recover_bev + first dilated
    convolution, sst_v2.py:139-197): the non-empty voxels are 41 % of the 468 x 468 cells on the bench frame, so the
    convolution runs as a sparse one - the cells with at least one voxel under their stencil are the output rows of a
    2-D rulebook (csrc/spconv.hip, dense-grid builder), the contraction and both gradients are the sparse-convolution
    kernels (csrc/spconv_os.hip: 9 x M pairs instead of 9 x all cells) - and only its OUTPUT is scattered into the dense
    canvas the batch norm behind it needs (every other cell is exactly 0, as in the dense convolution without bias)."""
    from . import spconv
    ny, nx = output_shape
    (kh, kw), (dh, dw), (ph, pw) = conv.kernel_size, conv.dilation, conv.padding
    zeros = torch.zeros_like(coors[:, 0])
    cells = torch.stack([coors[:, 0], zeros, coors[:, 2], coors[:, 3]], 1).int()
    outids, pairs, num = spconv.get_indice_pairs(cells, int(batch_size), [1, int(ny), int(nx)], [1, kh, kw], [1, 1, 1],
                                                 [0, ph, pw], [1, dh, dw], subm=False)
    # Conv2d weight [Cout, Cin, kh, kw] (cross-correlation) -> spconv filters [1, kh, kw, Cin, Cout]
    filters = conv.weight.permute(2, 3, 1, 0).reshape(1, kh, kw, conv.in_channels, conv.out_channels)
    rows = spconv.SparseConvFunction.apply(voxel_feat.contiguous(), filters, pairs, num, outids.size(0))
    return recover_bev(rows, outids.long(), batch_size, output_shape)


def _batch_size_of(info, coors_key):
    """number of samples: the input layer's own record when it left one, else the last batch index (one read-back, as
    in the reference: sst_v2.py:136)"""
    if info.get('batch_size') is not None:
        return int(info['batch_size'])
    coors = info[coors_key]
    return int(coors[:, 0].max().item()) + 1 if coors.size(0) > 0 else 1


class _WindowTransformer(_lib.Fp32Master, nn.Module):
    """What SSTv1 and SSTv2 have in common: optional input projection, the shift blocks, the attached convolutions."""

    def _build_stem(self, d_model, nhead, num_blocks, dim_feedforward, dropout, activation, in_channel, layer_cfg,
                    init_skip):
        if in_channel is not None:
            self.linear0 = nn.Linear(in_channel, d_model[0])
        self.block_list = nn.ModuleList([
            BasicShiftBlockV2(d_model[i], nhead[i], dim_feedforward[i], dropout, activation, batch_first=False, block_id=i,
                              layer_cfg=layer_cfg) for i in range(num_blocks)])
        # sst_v2.py:156-159 / sst_v1.py:216-219: Xavier on every matrix except the attention temperatures / scalers
        for name, p in self.named_parameters():
            if p.dim() > 1 and not any(tag in name for tag in init_skip):
                nn.init.xavier_uniform_(p)

    def _build_attached_convs(self, count, conv_in_channel, conv_out_channel, norm_cfg, conv_cfg, conv_kwargs):
        self.num_attached_conv = count
        if count <= 0:
            return
        if isinstance(conv_kwargs, (list, tuple)):
            assert len(conv_kwargs) == count
            per_layer = list(conv_kwargs)
        else:
            per_layer = [conv_kwargs] * count
        stages, width = [], conv_in_channel
        for kw in per_layer:
            stage = [build_conv_layer(conv_cfg, in_channels=width, out_channels=conv_out_channel, **kw)]
            if norm_cfg is not None:
                stage.append(build_norm_layer(norm_cfg, conv_out_channel)[1])
            stage.append(nn.ReLU(inplace=True))
            stages.append(nn.Sequential(*stage))
            width = conv_out_channel
        self.conv_layer = nn.ModuleList(stages)

    # the module's own arithmetic (class defaults = what a model built from a shipped config runs, no call needed): fp32 storage,
    # the dense products of the encoder layers from the exact three-way bf16 split (admissible as exact fp32:
    # tests/test_gpu_dense_f32x6.py).  set_precision('fp32') is the opt-out onto the fp32 matrix pipe.
    precision = 'fp32'
    matmul = 'f32x6'

    def set_precision(self, precision):
        """How THIS module's encoder layers compute (per module: nothing process-wide changes, two models of one process keep
        their own modes):
          'f32x6' (default) fp32 tensors and results (to fp32 rounding); the projections / FFN products from an EXACT three-way
                  bf16 split of both operands, six products with fp32 accumulation (csrc/dense_f32x6.hip): same arithmetic class
                  as exact fp32 (error vs float64 <= 2 x the fp32 matrix pipe's), 2.7 x less matrix-pipe time;
          'fp32'  every product on the fp32 matrix pipe (csrc/dense_f32.hip): the opt-out;
          'f32x3' two-way split, three products (~1e-5 relative, tighter than the TF32 the reference's torch 1.8 used on Ampere):
                  a measurement leg, never a default;
          'bf16'  reduced-precision encoder layers (sst_amd/bf16.py) - what the reference's fp16 training (Fp16OptimizerHook)
                  corresponds to on this hardware (cosine-attention layers included).  Layers the bf16 kernels do not cover (batch-norm layers, pre-norm, widths other than 128 / 256) keep
                  running in fp32 (split products)."""
        if precision not in ('fp32', 'bf16', 'f32x3', 'f32x6'):
            raise ValueError(precision)
        if precision == 'bf16':
            self.precision = 'bf16'     # self.matmul stays: the mode of whatever falls back to the fp32-storage layers
        else:
            self.precision = 'fp32'
            self.matmul = 'f32' if precision == 'fp32' else precision
        return self

    def run_blocks(self, feats, pos, plans, masks=None, pos_lookup=None):
        x = self.linear0(feats) if hasattr(self, 'linear0') else feats
        if self.precision == 'bf16' and pos_lookup is not None and x.is_cuda and x.dtype == torch.float32:
            from . import bf16
            layers = [enc for block in self.block_list for enc in block.encoder_list]
            if all(bf16.layer_supported(enc, plans[i % 2], x.size(0)) for i, enc in enumerate(layers)):
                return bf16.run_encoder_stack(self.block_list, x, plans, pos_lookup)
        if (self.precision == 'fp32' and pos_lookup is not None and x.is_cuda and x.dtype == torch.float32
                and x.size(1) == 128):
            # fp32 chain: every layer hands (x, x + positional embedding) to the next one (sst_basic_block.py); blocks listed
            # in checkpoint_blocks (sst_v2.py:131-133, torch.utils.checkpoint per block when training) are recomputed INSIDE
            # the chain instead of sending all layers to the per-layer path
            from .sst_basic_block import run_encoder_stack_fp32
            layers = [enc for block in self.block_list for enc in block.encoder_list]
            if all(enc._can_fuse(x, None, plans[i % 2]) for i, enc in enumerate(layers)):
                ckpt = self.checkpoint_blocks if self.training else ()
                return run_encoder_stack_fp32(self.block_list, x, plans, pos_lookup, checkpoint_blocks=ckpt)
        if pos_lookup is not None and any(p is None for p in pos):
            # the caller skipped the [M, C] positional tensors (frame plan with want_pos_rows = False): form them here
            pos = [t.index_select(0, idx.long()) for t, idx in pos_lookup]
        for i, block in enumerate(self.block_list):
            x = block(x, pos, plans, masks, using_checkpoint=i in self.checkpoint_blocks)
        return x

    def run_attached_convs(self, canvas, shortcut=False, first=0):
        for stage in list(getattr(self, 'conv_layer', ()))[first:]:
            y = stage(canvas)
            canvas = y + canvas if (shortcut and y.shape == canvas.shape) else y
        return canvas

    def bev_and_attached_convs(self, feats, coors, batch_size, shortcut=False):
        """recover_bev + the attached convolutions; the first convolution consumes the voxel rows directly when it can
        (sparse_first_conv) - no dense input canvas - unless a shortcut needs that canvas"""
        stages = getattr(self, 'conv_layer', ())
        if len(stages) > 0 and not shortcut and _sparse_first_conv_ok(stages[0][0], feats):
            y = sparse_first_conv(feats, coors, batch_size, self.output_shape, stages[0][0])
            for layer in list(stages[0])[1:]:     # norm, ReLU of the first stage
                y = layer(y)
            return self.run_attached_convs(y, shortcut, first=1)
        canvas = recover_bev(feats, coors, batch_size, self.output_shape)
        return self.run_attached_convs(canvas, shortcut)

    def set_impl(self, impl):
        """0: MFMA SRA kernels (default); 1: generic VALU kernels (in-library cross-check)."""
        for block in self.block_list:
            for enc in block.encoder_list:
                enc.win_attn.impl = impl

    def set_fused(self, fused):
        """True (default): each encoder layer is one autograd node; False: modular path (same arithmetic)."""
        for block in self.block_list:
            for enc in block.encoder_list:
                enc.fused = fused


@BACKBONES.register_module()
class SSTv2(_WindowTransformer):
    '''Single-stride Sparse Transformer (sst_v2.py:16-159).'''

    def __init__(self, d_model=[], nhead=[], num_blocks=6, dim_feedforward=[], dropout=0.0, activation="gelu",
                 output_shape=None, num_attached_conv=2, conv_in_channel=64, conv_out_channel=64,
                 norm_cfg=dict(type='naiveSyncBN2d', eps=1e-3, momentum=0.01), conv_cfg=dict(type='Conv2d', bias=False),
                 debug=True, in_channel=None, to_bev=True,
                 conv_kwargs=dict(kernel_size=3, dilation=2, padding=2, stride=1), checkpoint_blocks=[],
                 layer_cfg=dict(), conv_shortcut=False):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.checkpoint_blocks = checkpoint_blocks
        self.conv_shortcut = conv_shortcut
        self.to_bev = to_bev
        self.output_shape = output_shape
        self.debug = debug
        self._build_stem(d_model, nhead, num_blocks, dim_feedforward, dropout, activation, in_channel, layer_cfg,
                         init_skip=('scaler', 'tau'))
        self._build_attached_convs(num_attached_conv, conv_in_channel, conv_out_channel, norm_cfg, conv_cfg, conv_kwargs)

    @staticmethod
    def _window_inputs(voxel_info, shifts=2):
        """(plans, positional tensors, key masks) per partition, from either kind of voxel_info"""
        if 'sra_plan_shift0' in voxel_info:   # produced by this package's input layer / frame plan
            # the [M, C] positional tensors are formed on demand (VoxelInfo): when the (table, row index) pair is there the
            # encoder chains never need them, and run_blocks() gathers them itself for the per-layer path
            lazy = hasattr(voxel_info, 'peek') and 'pos_table' in voxel_info and 'pos_index_shift0' in voxel_info
            read = voxel_info.peek if lazy else voxel_info.get
            return ([voxel_info[f'sra_plan_shift{i}'] for i in range(shifts)],
                    [read(f'pos_embed_shift{i}') for i in range(shifts)], None)
        # the reference's per-level dictionaries (flat2win indices, padded positional tensors, key masks)
        return ([voxel_info[f'flat2win_inds_shift{i}'] for i in range(shifts)],
                [voxel_info[f'pos_dict_shift{i}'] for i in range(shifts)],
                [voxel_info[f'key_mask_shift{i}'] for i in range(shifts)])

    def forward_voxels(self, voxel_info):
        """the shift blocks only: [M', C] features of the kept voxels (what ``forward`` returns with ``to_bev=False``)"""
        feats_in = _lib.as_fp32(voxel_info['voxel_feats'])
        plans, pos, masks = self._window_inputs(voxel_info)
        lookup = None
        if 'pos_table' in voxel_info and 'pos_index_shift0' in voxel_info:   # (table, row index) per partition
            lookup = [(voxel_info['pos_table'], voxel_info[f'pos_index_shift{i}']) for i in range(2)]
        from . import dense
        # the reference's fp16 mode (wrap_fp16_model: model.half() + fp16_enabled on the encoder layers, whose auto_fp16 casts
        # their input to half: sst_basic_block_v2.py:102-104) = this stack's bf16 mode for the call; the features leave as
        # float16, as the reference's half layers hand them on (_lib.Fp32Master)
        half = _lib.wants_half(self)
        keep = self.precision
        if half:
            self.precision = 'bf16'
        try:
            with dense.matmul_mode_scope(getattr(self, 'matmul', None)):     # this module's own mode, whatever another model set
                out = self.run_blocks(feats_in, pos, plans, masks, pos_lookup=lookup)
        finally:
            self.precision = keep
        return out.half() if half else out

    def forward(self, voxel_info):
        coors = voxel_info['voxel_coors']
        assert coors.dtype == torch.int64, 'data type of coors should be torch.int64!'
        feats = self.forward_voxels(voxel_info)
        half = _lib.wants_half(self)
        if not self.to_bev:
            assert self.num_attached_conv <= 0, 'the attached convolutions need the BEV canvas'
            return [{'voxel_feats': feats, 'voxel_coors': coors}]
        # the canvas and the attached convolutions compute in fp32 (their parameters are the fp32 masters)
        bev = self.bev_and_attached_convs(_lib.as_fp32(feats), coors, _batch_size_of(voxel_info, 'voxel_coors'), self.conv_shortcut)
        return [bev.half() if half else bev]      # fp16 mode: the neck / heads behind are half modules

    def recover_bev(self, voxel_feat, coors, batch_size):
        return recover_bev(voxel_feat, coors, batch_size, self.output_shape)


@BACKBONES.register_module()
class SSTv1(_WindowTransformer):
    """First-generation backbone (mmdet3d/models/backbones/sst_v1.py:17-270): consumes the 3-tuple of SSTInputLayer,
    computes the positional embedding itself (:221-259), always recovers the BEV canvas.  The encoder layers are the
    same modules as SSTv2's (the v1 blocks have identical parameters, mmdet3d/models/sst/sst_basic_block.py:62-99)."""

    def __init__(self, d_model=[], nhead=[], num_blocks=6, dim_feedforward=[], dropout=0.0, activation="gelu",
                 output_shape=None, num_attached_conv=2, conv_in_channel=64, conv_out_channel=64,
                 norm_cfg=dict(type='naiveSyncBN2d', eps=1e-3, momentum=0.01),
                 conv_cfg=dict(type='Conv2d', bias=False), debug=True, drop_info=None, normalize_pos=False,
                 pos_temperature=10000, window_shape=None, in_channel=None,
                 conv_kwargs=dict(kernel_size=3, dilation=2, padding=2, stride=1), checkpoint_blocks=[]):
        super().__init__()
        assert drop_info is not None
        self.meta_drop_info = drop_info
        self.pos_temperature, self.normalize_pos = pos_temperature, normalize_pos
        self.d_model, self.nhead, self.window_shape = d_model, nhead, window_shape
        self.checkpoint_blocks = checkpoint_blocks
        self.output_shape = output_shape
        self.debug = debug
        self._build_stem(d_model, nhead, num_blocks, dim_feedforward, dropout, activation, in_channel, dict(),
                         init_skip=('scaler',))
        self._build_attached_convs(num_attached_conv, conv_in_channel, conv_out_channel, norm_cfg, conv_cfg, conv_kwargs)

    def set_drop_info(self):
        if not hasattr(self, 'drop_info'):
            meta = self.meta_drop_info
            self.drop_info = (meta[0] if self.training else meta[1]) if isinstance(meta, tuple) else meta

    @torch.no_grad()
    def get_pos_embed_flat(self, coors_in_win, dtype):
        """[M, d_model] embedding from v1 in-window coordinates (x, y); arithmetic of sst_v1.py:221-259: per axis
        d_model / 2 values, sin on the even and cos on the odd entries of centred_coordinate / T^(2 (i // 2) / L)."""
        half = self.d_model[0] // 2
        idx = torch.arange(half, dtype=torch.float32, device=coors_in_win.device)
        denom = self.pos_temperature ** (2 * (idx // 2) / half)
        parts = []
        for axis, extent in enumerate(self.window_shape):
            centred = coors_in_win[:, axis] - extent / 2
            if self.normalize_pos:
                centred = centred / extent * 2 * 3.1415
            phase = centred[:, None] / denom[None, :]
            parts.append(torch.stack([phase[:, ::2].sin(), phase[:, 1::2].cos()], dim=-1).flatten(1))
        return torch.cat(parts, dim=-1).to(dtype)

    def forward(self, input_tuple):
        voxel_feat, ind_dict_list, voxel_info = input_tuple
        voxel_feat = _lib.as_fp32(voxel_feat)
        assert voxel_info['coors'].dtype == torch.int64, 'data type of coors should be torch.int64!'
        self.set_drop_info()
        plans, pos = [], []
        for i, ind_dict in enumerate(ind_dict_list):
            plan = voxel_info.get(f'sra_plan_shift{i}')
            if plan is None:   # a voxel_info produced by the reference's own SSTInputLayer: convert its dictionaries
                plan = plan_from_reference_dicts(dict(ind_dict, batching_info=self.drop_info), voxel_feat.size(0),
                                                 voxel_feat.device)
            plans.append(plan)
            pos.append(self.get_pos_embed_flat(voxel_info[f'coors_in_win_shift{i}'], voxel_feat.dtype))
        from . import dense
        with dense.matmul_mode_scope(getattr(self, 'matmul', None)):
            feats = self.run_blocks(voxel_feat, pos, plans)
        return [self.bev_and_attached_convs(feats, voxel_info['coors'], _batch_size_of(voxel_info, 'coors'))]


@BACKBONES.register_module()
class SIR(nn.Module):
    '''Sparse Instance Recognition backbone: a stack of SIRLayer (sir.py:15-87).'''

    def __init__(self, num_blocks=5, in_channels=[], feat_channels=[], rel_mlp_hidden_dims=[], with_rel_mlp=True,
                 with_distance=False, with_cluster_center=False, norm_cfg=dict(type='LN', eps=1e-3), mode='max',
                 xyz_normalizer=[1.0, 1.0, 1.0], act='relu', dropout=0, unique_once=False):
        super().__init__()
        self.num_blocks = num_blocks
        self.unique_once = unique_once
        common = dict(type='SIRLayer', with_distance=with_distance, with_cluster_center=with_cluster_center,
                      with_rel_mlp=with_rel_mlp, with_voxel_center=False, norm_cfg=norm_cfg, mode=mode, fusion_layer=None,
                      return_inv=False, rel_dist_scaler=10.0, xyz_normalizer=xyz_normalizer, act=act, dropout=dropout,
                      # placeholders the layer never reads (sir.py:47-48)
                      voxel_size=[0.1, 0.1, 0.1], point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4])
        self.block_list = nn.ModuleList([
            build_voxel_encoder(dict(common, in_channels=in_channels[i], feat_channels=feat_channels[i],
                                     rel_mlp_hidden_dims=rel_mlp_hidden_dims[i],
                                     return_point_feats=i != num_blocks - 1)) for i in range(num_blocks)])

    def forward(self, points, features, coors, f_cluster=None):
        """-> (point features of the last block, per-cluster features of all blocks side by side, cluster coordinates)."""
        features = _lib.as_fp32(features)
        grouping = dict(new_coors_once=None, unq_inv_once=None)
        if self.unique_once:   # one sorted-unique of the cluster ids for the whole stack (its CSR rides on unq_inv)
            grouping['new_coors_once'], grouping['unq_inv_once'] = unique_with_plan(coors)
        point_feats, per_cluster, cluster_coors = features, [], None
        last = self.num_blocks - 1
        for i, block in enumerate(self.block_list):
            block_in = torch.cat([points, point_feats], 1)
            if i < last:
                point_feats, cluster_feats = block(block_in, coors, f_cluster, **grouping)
            else:
                point_feats, cluster_feats, cluster_coors = block(block_in, coors, f_cluster, return_both=True, **grouping)
            per_cluster.append(cluster_feats)
        return point_feats, torch.cat(per_cluster, dim=1), cluster_coors
