"""SSTv2 and SIR backbones.

Mirrors mmdet3d/models/backbones/sst_v2.py:16-196 and mmdet3d/models/backbones/sir.py:15-87 (registry
names 'SSTv2' / 'SIR', constructor kwargs, forward signatures, state_dict keys
``block_list.{i}.encoder_list.{0,1}.*``, ``conv_layer.{j}.{0,1}.*``, ``linear0.*``).
"""
import torch
import torch.nn as nn

from .norm import build_conv_layer, build_norm_layer
from .registry import BACKBONES, build_voxel_encoder
from .sst_basic_block import BasicShiftBlockV2, plan_from_reference_dicts
from .sst_ops import unique_with_plan


@BACKBONES.register_module()
class SSTv2(nn.Module):
    '''Single-stride Sparse Transformer (sst_v2.py:16-159).'''

    def __init__(
        self,
        d_model=[],
        nhead=[],
        num_blocks=6,
        dim_feedforward=[],
        dropout=0.0,
        activation="gelu",
        output_shape=None,
        num_attached_conv=2,
        conv_in_channel=64,
        conv_out_channel=64,
        norm_cfg=dict(type='naiveSyncBN2d', eps=1e-3, momentum=0.01),
        conv_cfg=dict(type='Conv2d', bias=False),
        debug=True,
        in_channel=None,
        to_bev=True,
        conv_kwargs=dict(kernel_size=3, dilation=2, padding=2, stride=1),
        checkpoint_blocks=[],
        layer_cfg=dict(),
        conv_shortcut=False,
    ):
        super().__init__()
        self.d_model = d_model
        self.nhead = nhead
        self.checkpoint_blocks = checkpoint_blocks
        self.conv_shortcut = conv_shortcut
        self.to_bev = to_bev

        if in_channel is not None:
            self.linear0 = nn.Linear(in_channel, d_model[0])

        block_list = []
        for i in range(num_blocks):
            block_list.append(
                BasicShiftBlockV2(d_model[i], nhead[i], dim_feedforward[i], dropout, activation, batch_first=False,
                                  block_id=i, layer_cfg=layer_cfg))
        self.block_list = nn.ModuleList(block_list)
        self._reset_parameters()
        self.output_shape = output_shape
        self.debug = debug
        self.num_attached_conv = num_attached_conv

        if num_attached_conv > 0:
            conv_list = []
            for i in range(num_attached_conv):
                if isinstance(conv_kwargs, dict):
                    conv_kwargs_i = conv_kwargs
                elif isinstance(conv_kwargs, list):
                    assert len(conv_kwargs) == num_attached_conv
                    conv_kwargs_i = conv_kwargs[i]
                if i > 0:
                    conv_in_channel = conv_out_channel
                conv = build_conv_layer(conv_cfg, in_channels=conv_in_channel, out_channels=conv_out_channel,
                                        **conv_kwargs_i)
                if norm_cfg is None:
                    convnormrelu = nn.Sequential(conv, nn.ReLU(inplace=True))
                else:
                    convnormrelu = nn.Sequential(conv, build_norm_layer(norm_cfg, conv_out_channel)[1],
                                                 nn.ReLU(inplace=True))
                conv_list.append(convnormrelu)
            self.conv_layer = nn.ModuleList(conv_list)

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

    def forward(self, voxel_info):
        num_shifts = 2
        assert voxel_info['voxel_coors'].dtype == torch.int64, 'data type of coors should be torch.int64!'
        voxel_feat = voxel_info['voxel_feats']
        if 'sra_plan_shift0' in voxel_info:   # produced by this package's SSTInputLayerV2
            ind_dict_list = [voxel_info[f'sra_plan_shift{i}'] for i in range(num_shifts)]
            pos_embed_list = [voxel_info[f'pos_embed_shift{i}'] for i in range(num_shifts)]
            padding_mask_list = None
        else:                                 # reference-style dictionaries
            ind_dict_list = [voxel_info[f'flat2win_inds_shift{i}'] for i in range(num_shifts)]
            padding_mask_list = [voxel_info[f'key_mask_shift{i}'] for i in range(num_shifts)]
            pos_embed_list = [voxel_info[f'pos_dict_shift{i}'] for i in range(num_shifts)]

        output = voxel_feat
        if hasattr(self, 'linear0'):
            output = self.linear0(output)
        for i, block in enumerate(self.block_list):
            output = block(output, pos_embed_list, ind_dict_list, padding_mask_list,
                           using_checkpoint=i in self.checkpoint_blocks)

        if self.to_bev:
            batch_size = voxel_info['voxel_coors'][:, 0].max().item() + 1
            output = self.recover_bev(output, voxel_info['voxel_coors'], batch_size)

        output_list = []
        if self.num_attached_conv > 0:
            assert self.to_bev
            for conv in self.conv_layer:
                temp = conv(output)
                if temp.shape == output.shape and self.conv_shortcut:
                    output = temp + output
                else:
                    output = temp

        if not self.to_bev:
            output = {'voxel_feats': output, 'voxel_coors': voxel_info['voxel_coors']}
        output_list.append(output)
        return output_list

    def _reset_parameters(self):
        for name, p in self.named_parameters():
            if p.dim() > 1 and 'scaler' not in name and 'tau' not in name:
                nn.init.xavier_uniform_(p)

    def recover_bev(self, voxel_feat, coors, batch_size):
        '''[N,C] voxel features -> dense [B, C, ny, nx] canvas (sst_v2.py:161-197), one scatter for the whole
        batch instead of a python loop; rows are written token-major (coalesced) and the result is returned
        as a channels-last view of logical shape [B,C,ny,nx].'''
        ny, nx = self.output_shape
        feat_dim = voxel_feat.shape[-1]
        canvas = voxel_feat.new_zeros((batch_size * ny * nx, feat_dim))
        flat = coors[:, 0] * (ny * nx) + coors[:, 2] * nx + coors[:, 3]
        canvas = canvas.index_put((flat.long(),), voxel_feat)
        return canvas.view(batch_size, ny, nx, feat_dim).permute(0, 3, 1, 2)


@BACKBONES.register_module()
class SSTv1(nn.Module):
    """First-generation backbone (mmdet3d/models/backbones/sst_v1.py:17-270): consumes the 3-tuple of
    SSTInputLayer, computes positional embedding itself (:221-259), always recovers the BEV canvas.
    Same constructor kwargs, forward signature and state_dict keys; the encoder layers are the same modules as
    SSTv2's (the v1 blocks have identical parameters, mmdet3d/models/sst/sst_basic_block.py:62-99)."""

    def __init__(self, d_model=[], nhead=[], num_blocks=6, dim_feedforward=[], dropout=0.0, activation="gelu",
                 output_shape=None, num_attached_conv=2, conv_in_channel=64, conv_out_channel=64,
                 norm_cfg=dict(type='naiveSyncBN2d', eps=1e-3, momentum=0.01),
                 conv_cfg=dict(type='Conv2d', bias=False), debug=True, drop_info=None, normalize_pos=False,
                 pos_temperature=10000, window_shape=None, in_channel=None,
                 conv_kwargs=dict(kernel_size=3, dilation=2, padding=2, stride=1), checkpoint_blocks=[]):
        super().__init__()
        assert drop_info is not None
        self.meta_drop_info = drop_info
        self.pos_temperature = pos_temperature
        self.d_model = d_model
        self.window_shape = window_shape
        self.normalize_pos = normalize_pos
        self.nhead = nhead
        self.checkpoint_blocks = checkpoint_blocks
        if in_channel is not None:
            self.linear0 = nn.Linear(in_channel, d_model[0])
        self.block_list = nn.ModuleList([
            BasicShiftBlockV2(d_model[i], nhead[i], dim_feedforward[i], dropout, activation, batch_first=False,
                              block_id=i) for i in range(num_blocks)])
        for name, p in self.named_parameters():
            if p.dim() > 1 and 'scaler' not in name:
                nn.init.xavier_uniform_(p)
        self.output_shape = output_shape
        self.debug = debug
        self.num_attached_conv = num_attached_conv
        if num_attached_conv > 0:
            conv_list = []
            for i in range(num_attached_conv):
                conv_kwargs_i = conv_kwargs if isinstance(conv_kwargs, dict) else conv_kwargs[i]
                if i > 0:
                    conv_in_channel = conv_out_channel
                conv = build_conv_layer(conv_cfg, in_channels=conv_in_channel, out_channels=conv_out_channel,
                                        **conv_kwargs_i)
                layers = [conv] if norm_cfg is None else [conv, build_norm_layer(norm_cfg, conv_out_channel)[1]]
                conv_list.append(nn.Sequential(*layers, nn.ReLU(inplace=True)))
            self.conv_layer = nn.ModuleList(conv_list)

    def set_drop_info(self):
        if hasattr(self, 'drop_info'):
            return
        meta = self.meta_drop_info
        if isinstance(meta, tuple):
            self.drop_info = meta[0] if self.training else meta[1]
        else:
            self.drop_info = meta

    @torch.no_grad()
    def get_pos_embed_flat(self, coors_in_win, dtype):
        """[M, d_model] embedding from v1 in-window coordinates (x, y); arithmetic of sst_v1.py:221-259."""
        win_x, win_y = self.window_shape
        x, y = coors_in_win[:, 0] - win_x / 2, coors_in_win[:, 1] - win_y / 2
        if self.normalize_pos:
            x = x / win_x * 2 * 3.1415
            y = y / win_y * 2 * 3.1415
        pos_length = self.d_model[0] // 2
        inv_freq = torch.arange(pos_length, dtype=torch.float32, device=coors_in_win.device)
        inv_freq = self.pos_temperature ** (2 * (inv_freq // 2) / pos_length)
        embed_x = x[:, None] / inv_freq[None, :]
        embed_y = y[:, None] / inv_freq[None, :]
        embed_x = torch.stack([embed_x[:, ::2].sin(), embed_x[:, 1::2].cos()], dim=-1).flatten(1)
        embed_y = torch.stack([embed_y[:, ::2].sin(), embed_y[:, 1::2].cos()], dim=-1).flatten(1)
        return torch.cat([embed_x, embed_y], dim=-1).to(dtype)

    def forward(self, input_tuple):
        voxel_feat, ind_dict_list, voxel_info = input_tuple
        assert voxel_info['coors'].dtype == torch.int64, 'data type of coors should be torch.int64!'
        self.set_drop_info()
        batch_size = voxel_info['coors'][:, 0].max().item() + 1
        num_shifts = len(ind_dict_list)
        m = voxel_feat.size(0)
        plans, pos = [], []
        for i in range(num_shifts):
            if f'sra_plan_shift{i}' in voxel_info:
                plans.append(voxel_info[f'sra_plan_shift{i}'])
            else:  # a voxel_info produced by the reference's own SSTInputLayer
                d = dict(ind_dict_list[i])
                d['batching_info'] = self.drop_info
                plans.append(plan_from_reference_dicts(d, m, voxel_feat.device))
            pos.append(self.get_pos_embed_flat(voxel_info[f'coors_in_win_shift{i}'], voxel_feat.dtype))
        output = voxel_feat
        if hasattr(self, 'linear0'):
            output = self.linear0(output)
        for i, block in enumerate(self.block_list):
            output = block(output, pos, plans, None, using_checkpoint=i in self.checkpoint_blocks)
        output = SSTv2.recover_bev(self, output, voxel_info['coors'], batch_size)
        if self.num_attached_conv > 0:
            for conv in self.conv_layer:
                output = conv(output)
        return [output]


@BACKBONES.register_module()
class SIR(nn.Module):
    '''Sparse Instance Recognition backbone: a stack of SIRLayer (sir.py:15-87).'''

    def __init__(
        self,
        num_blocks=5,
        in_channels=[],
        feat_channels=[],
        rel_mlp_hidden_dims=[],
        with_rel_mlp=True,
        with_distance=False,
        with_cluster_center=False,
        norm_cfg=dict(type='LN', eps=1e-3),
        mode='max',
        xyz_normalizer=[1.0, 1.0, 1.0],
        act='relu',
        dropout=0,
        unique_once=False,
    ):
        super().__init__()
        self.num_blocks = num_blocks
        self.unique_once = unique_once
        block_list = []
        for i in range(num_blocks):
            return_point_feats = i != num_blocks - 1
            kwargs = dict(
                type='SIRLayer',
                in_channels=in_channels[i],
                feat_channels=feat_channels[i],
                with_distance=with_distance,
                with_cluster_center=with_cluster_center,
                with_rel_mlp=with_rel_mlp,
                rel_mlp_hidden_dims=rel_mlp_hidden_dims[i],
                with_voxel_center=False,
                voxel_size=[0.1, 0.1, 0.1],  # not used, placeholder
                point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],  # not used, placeholder
                norm_cfg=norm_cfg,
                mode=mode,
                fusion_layer=None,
                return_point_feats=return_point_feats,
                return_inv=False,
                rel_dist_scaler=10.0,
                xyz_normalizer=xyz_normalizer,
                act=act,
                dropout=dropout,
            )
            block_list.append(build_voxel_encoder(kwargs))
        self.block_list = nn.ModuleList(block_list)

    def forward(self, points, features, coors, f_cluster=None):
        if self.unique_once:
            new_coors, unq_inv = unique_with_plan(coors)
        else:
            new_coors = unq_inv = None
        out_feats = features
        cluster_feat_list = []
        for i, block in enumerate(self.block_list):
            in_feats = torch.cat([points, out_feats], 1)
            if i < self.num_blocks - 1:
                out_feats, out_cluster_feats = block(in_feats, coors, f_cluster, unq_inv_once=unq_inv,
                                                     new_coors_once=new_coors)
                cluster_feat_list.append(out_cluster_feats)
            if i == self.num_blocks - 1:
                out_feats, out_cluster_feats, out_coors = block(in_feats, coors, f_cluster, return_both=True,
                                                                unq_inv_once=unq_inv, new_coors_once=new_coors)
                cluster_feat_list.append(out_cluster_feats)
        final_cluster_feats = torch.cat(cluster_feat_list, dim=1)
        return out_feats, final_cluster_feats, out_coors
