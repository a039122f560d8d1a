"""sst_amd — MI355X-native (gfx950) hot path of the SST / FSD sparse LiDAR backbones.

The names exported here are the ones the reference exposes from ``mmdet3d.ops`` / its registries for the
path; they are backed by hand-written HIP kernels in ``sst_amd/csrc`` behind the C ABI of
``include/sst_amd.h``.  There is no CPU or eager fallback: without the built library, or on a CPU
tensor, the ops raise.
"""
from . import _lib
from .kernels import WindowPlan, sra_attention, sra_attention_qk_v
from .norm import NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d, build_conv_layer, build_norm_layer
from .registry import (BACKBONES, MIDDLE_ENCODERS, MODELS, ROI_EXTRACTORS, VOXEL_ENCODERS, build_backbone,
                       build_middle_encoder, build_voxel_encoder)
from .voxel import (DynamicScatter, Voxelization, build_scatter_plan, dynamic_point_to_voxel_forward,
                    dynamic_scatter, dynamic_voxelize, voxelization)
from .sst_ops import (build_mlp, flat2window, flat2window_v2, get_activation, get_activation_layer,
                      get_flat2win_inds, get_flat2win_inds_v2, get_inner_win_inds, get_window_coors,
                      make_continuous_inds, scatter_v2, window2flat, window2flat_v2)
from .voxel_encoder import DynamicScatterVFE, DynamicVFE, DynamicVFELayer, DynamicVFELayerV2, SIRLayer
from .sst_input_layer import PseudoMiddleEncoderForSpconvFSD, SSTInputLayer, SSTInputLayerV2
from .sst_basic_block import BasicShiftBlockV2, EncoderLayer, WindowAttention
from .backbones import SIR, SSTv1, SSTv2
from .cluster import (ClusterAssigner, connected_components_xy, filter_almost_empty, find_connected_componets,  # noqa: F401
                      find_connected_componets_single_batch, modify_cluster_by_class)
from .dynamic_point_pool import DynamicPointROIExtractor, dynamic_point_pool, dynamic_point_pool_mixed
from . import spconv  # noqa: F401  (sst_amd.spconv mirrors mmdet3d.ops.spconv)
from .spconv import (SparseConv3d, SparseConvTensor, SparseConvTranspose3d, SparseInverseConv3d,  # noqa: F401
                     SparseMaxPool3d, SparseModule, SparseSequential, SubMConv3d)
from .sparse_unet import (SimpleSparseUNet, SparseBasicBlock, SparseUNet, VirtualVoxelMixer,  # noqa: F401
                          make_sparse_convmodule)

from .virtual_voxel import VirtualVoxelExtractor  # noqa: F401
from . import detectors  # noqa: F401
from .detectors import (DETECTORS, HEADS, NECKS, FSD, FSDV2, DynamicCenterPoint, DynamicVoxelNet, SingleStageFSD, SingleStageFSDV2, VoteSegHead,  # noqa: F401
                        VoteSegmentor, Voxel2PointScatterNeck, build_detector, build_head, build_model, build_neck,
                        install_fused_extract_feat)

__version__ = '0.1.0'

__all__ = [
    'VirtualVoxelExtractor', 'detectors', 'DynamicVoxelNet', 'DynamicCenterPoint', 'install_fused_extract_feat', 'DETECTORS', 'HEADS', 'NECKS', 'FSD', 'FSDV2', 'SingleStageFSD', 'SingleStageFSDV2',
    'VoteSegHead', 'VoteSegmentor', 'Voxel2PointScatterNeck', 'build_detector', 'build_head', 'build_model', 'build_neck',
    'Voxelization', 'voxelization', 'DynamicScatter', 'dynamic_scatter', 'dynamic_voxelize',
    'dynamic_point_to_voxel_forward', 'build_scatter_plan', 'flat2window', 'window2flat', 'get_flat2win_inds',
    'get_inner_win_inds', 'make_continuous_inds', 'flat2window_v2', 'window2flat_v2', 'get_flat2win_inds_v2',
    'get_window_coors', 'scatter_v2', 'build_mlp', 'get_activation', 'get_activation_layer',
    'NaiveSyncBatchNorm1d', 'NaiveSyncBatchNorm2d', 'build_norm_layer', 'build_conv_layer', 'DynamicVFE',
    'DynamicScatterVFE', 'SIRLayer', 'DynamicVFELayer', 'DynamicVFELayerV2', 'SSTInputLayer', 'SSTInputLayerV2',
    'PseudoMiddleEncoderForSpconvFSD', 'WindowAttention', 'EncoderLayer', 'BasicShiftBlockV2', 'SSTv1', 'SSTv2', 'SIR',
    'MODELS', 'VOXEL_ENCODERS', 'MIDDLE_ENCODERS', 'BACKBONES', 'build_voxel_encoder', 'build_middle_encoder',
    'build_backbone', 'WindowPlan', 'sra_attention', 'sra_attention_qk_v', 'ClusterAssigner',
    'find_connected_componets', 'find_connected_componets_single_batch', 'filter_almost_empty',
    'modify_cluster_by_class', 'connected_components_xy', 'ROI_EXTRACTORS', 'DynamicPointROIExtractor',
    'dynamic_point_pool', 'dynamic_point_pool_mixed', 'spconv', 'SparseConvTensor', 'SparseSequential', 'SparseModule',
    'SubMConv3d', 'SparseConv3d', 'SparseConvTranspose3d', 'SparseInverseConv3d', 'SparseMaxPool3d', 'SparseUNet', 'SimpleSparseUNet', 'VirtualVoxelMixer',
    'SparseBasicBlock', 'make_sparse_convmodule',
]
