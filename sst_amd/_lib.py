"""ctypes binding of libsst_amd.so (the C ABI declared in include/sst_amd.h).

PyTorch is used here only as the owner of device memory and of the HIP stream: every entry point
receives raw device pointers (``tensor.data_ptr()``), sizes and ``torch.cuda.current_stream()``.
There is no CPU fallback: if the library is missing, or a tensor is not on the GPU, the call raises.
"""
import ctypes
import os
import subprocess

import torch  # must be imported before the CDLL so that libamdhip64.so.7 resolves to torch's copy

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_CSRC, 'libsst_amd.so')

_lib = None

c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p



class WgradProblemBF16(ctypes.Structure):
    """sst_wgrad_problem_bf16 of include/sst_amd.h"""
    _fields_ = [('a', c_ptr), ('b', c_ptr), ('lda', c_i64), ('ldb', c_i64), ('m', c_i64), ('out_w', c_ptr),
                ('out_b', c_ptr), ('p', ctypes.c_int32), ('bias_side', ctypes.c_int32),
                ('transpose_out', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class CastProblemBF16(ctypes.Structure):
    """sst_cast_problem_bf16 of include/sst_amd.h"""
    _fields_ = [('src', c_ptr), ('dst', c_ptr), ('ld_src', c_i64), ('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
                ('transpose', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class WgradProblemF32(ctypes.Structure):
    """sst_wgrad_problem_f32 of include/sst_amd.h"""
    _fields_ = [('dy', c_ptr), ('x', c_ptr), ('m', c_i64), ('ld_dy', c_i64), ('ld_x', c_i64), ('dw', c_ptr), ('db', c_ptr),
                ('out', ctypes.c_int32), ('inn', ctypes.c_int32), ('x_add_rows', c_ptr), ('x_add_index', c_ptr)]


def _struct(name, doc, fields):
    return type(name, (ctypes.Structure,), {'__doc__': doc, '_fields_': fields})


_P = ctypes.c_void_p
EncoderLayerFwdArgs = _struct('EncoderLayerFwdArgs', 'sst_encoder_layer_fwd_args of include/sst_amd.h', (
    [('m', c_i64), ('n_windows', c_i64)] + [(k, ctypes.c_int32) for k in ('n_heads', 'act', 'max_tokens', 'impl')]
    + [('eps', ctypes.c_float), ('scale', ctypes.c_float)]
    + [(k, _P) for k in ('x', 'xp', 'w_in', 'b_in', 'w_out', 'b_out', 'w1', 'b1', 'w2', 'b2', 'n1w', 'n1b', 'n2w', 'n2b',
                         'tok', 'winoff', 'order', 'pos_table', 'pos_idx',
                         'qkv', 'o', 'lse', 'y1', 's1', 'st1', 'pre', 'h', 's2', 'y2', 'st2', 'y2p', 'head_scale', 'wpack',
                         'xpos_table', 'xpos_idx')]))
EncoderLayerBwdArgs = _struct('EncoderLayerBwdArgs', 'sst_encoder_layer_bwd_args of include/sst_amd.h', (
    [('m', c_i64), ('n_windows', c_i64)] + [(k, ctypes.c_int32) for k in ('n_heads', 'act', 'max_tokens', 'impl')]
    + [('eps', ctypes.c_float), ('scale', ctypes.c_float)]
    + [(k, _P) for k in ('dy2', 'dy2p', 'x', 'xp', 'qkv', 'o', 'lse', 's1', 'st1', 'y1', 'pre', 'h', 's2', 'st2',
                         'w_in', 'w_out', 'w1', 'w2', 'n1w', 'n2w', 'tok', 'winoff', 'order',
                         'ds2', 'dpre', 'ds1', 'd_o', 'dqkv',
                         'dw_in', 'db_in', 'dwo', 'dbo', 'dw1', 'db1', 'dw2', 'db2', 'dn1w', 'dn1b', 'dn2w', 'dn2b',
                         'workspace', 'head_scale', 'cos_r', 'dy1', 'wpack', 'xpos_table', 'xpos_idx')]))


EncoderTailFwdArgs = _struct('EncoderTailFwdArgs', 'sst_encoder_tail_fwd_args of include/sst_amd.h', (
    [('m', c_i64), ('act', ctypes.c_int32), ('reserved', ctypes.c_int32), ('eps', ctypes.c_float), ('reserved_f', ctypes.c_float)]
    + [(k, _P) for k in ('o', 'x', 'packed', 'b_out', 'b1', 'b2', 'n1w', 'n1b', 'n2w', 'n2b', 'pos_table', 'pos_idx',
                         's1', 'st1', 'y1', 'pre', 'h', 's2', 'st2', 'y2', 'y2p')]))
EncoderTailBwdArgs = _struct('EncoderTailBwdArgs', 'sst_encoder_tail_bwd_args of include/sst_amd.h', (
    [('m', c_i64), ('act', ctypes.c_int32), ('reserved', ctypes.c_int32)]
    + [(k, _P) for k in ('dy2', 'dy2p', 's2', 'st2', 'pre', 's1', 'st1', 'packed', 'n1w', 'n2w',
                         'ds2', 'dpre', 'ds1', 'd_o', 'dn2w', 'dn2b', 'dn1w', 'dn1b', 'workspace')]))

EncoderTailFwdBF16Args = _struct('EncoderTailFwdBF16Args', 'sst_encoder_tail_fwd_bf16_args of include/sst_amd.h', (
    [('m', c_i64), ('act', ctypes.c_int32), ('reserved', ctypes.c_int32), ('eps', ctypes.c_float), ('reserved_f', ctypes.c_float)]
    + [(k, _P) for k in ('o', 'x', 'packed', 'b_out', 'b1', 'b2', 'n1w', 'n1b', 'n2w', 'n2b', 'pos_table', 'pos_idx',
                         's1', 'st1', 'y1', 'pre', 'h', 's2', 'st2', 'y2', 'y2p')]))
EncoderTailBwdBF16Args = _struct('EncoderTailBwdBF16Args', 'sst_encoder_tail_bwd_bf16_args of include/sst_amd.h', (
    [('m', c_i64), ('act', ctypes.c_int32), ('reserved', ctypes.c_int32)]
    + [(k, _P) for k in ('dy2', 'dy2p', 's2', 'st2', 'pre', 's1', 'st1', 'packed', 'n1w', 'n2w',
                         'ds2', 'dpre', 'ds1', 'd_o', 'dn2w', 'dn2b', 'dn1w', 'dn1b', 'workspace')]))


EncoderLayerFwdBF16Args = _struct('EncoderLayerFwdBF16Args', 'sst_encoder_layer_fwd_bf16_args of include/sst_amd.h', (
    [('m', c_i64), ('n_windows', c_i64)] + [(k, ctypes.c_int32) for k in ('n_heads', 'act', 'max_tokens', 'reserved')]
    + [('eps', ctypes.c_float), ('scale', ctypes.c_float)]
    + [(k, _P) for k in ('x', 'xp', 'wqk', 'wv', 'wout', 'w1', 'w2', 'b_in', 'b_out', 'b1', 'b2', 'n1w', 'n1b', 'n2w', 'n2b',
                         'tok', 'winoff', 'order', 'pos_table', 'pos_idx',
                         'qk', 'v', 'o', 'lse', 'y1', 's1', 'st1', 'pre', 'h', 's2', 'st2', 'y2', 'y2p', 'head_scale', 'wpack')]))
EncoderLayerBwdBF16Args = _struct('EncoderLayerBwdBF16Args', 'sst_encoder_layer_bwd_bf16_args of include/sst_amd.h', (
    [('m', c_i64), ('n_windows', c_i64)] + [(k, ctypes.c_int32) for k in ('n_heads', 'act', 'max_tokens', 'reserved')]
    + [('eps', ctypes.c_float), ('scale', ctypes.c_float)]
    + [(k, _P) for k in ('dy2', 'dy2p', 'x', 'xp', 'qk', 'v', 'o', 'lse', 's1', 'st1', 'y1', 'pre', 'h', 's2', 'st2',
                         'wqk_t', 'wv_t', 'wout_t', 'w1_t', 'w2_t', 'n1w', 'n2w', 'tok', 'winoff', 'order',
                         'ds2', 'dpre', 'dy1', 'ds1', 'd_o', 'dqkv', 'dxp', 'dx',
                         'dw_in', 'db_in', 'dwo', 'dbo', 'dw1', 'db1', 'dw2', 'db2', 'dn1w', 'dn1b', 'dn2w', 'dn2b',
                         'workspace', 'head_scale', 'cos_r', 'wpack')]))


# name -> (restype, argtypes); mirrors include/sst_amd.h one to one
_SIGNATURES = {
    'sst_version': (ctypes.c_char_p, []),
    'sst_dynamic_voxelize_f32': (c_i32, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr]),
    'sst_dynamic_voxelize_grid': (None, [c_ptr, c_ptr, c_ptr]),
    'sst_scan_workspace_bytes': (c_i64, [c_i64]),
    'sst_exclusive_scan_i32': (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    'sst_sort_workspace_bytes': (c_i64, [c_i64]),
    'sst_sort_pairs_u64': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    'sst_unique_workspace_bytes': (c_i64, [c_i64]),
    'sst_unique_rows': (c_i32, [c_ptr, c_i32, c_i64, c_i32, c_i64, c_ptr, c_ptr, c_i32,
                                c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_unpack_keys': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_i64, c_i32, c_ptr]),
    'sst_segment_reduce_fwd_f32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr,
                                           c_ptr, c_ptr]),
    'sst_encoder_layer_bwd_workspace_bytes': (c_i64, [c_i64, c_i32]),
    'sst_inproj_pos_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'sst_encoder_layer_wpack_bytes': (c_i64, []),
    'sst_encoder_layer_fwd_f32x6': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_layer_bwd_f32x6': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_tail_pack_bytes': (c_i64, []),
    'sst_encoder_tail_pack_f32x6': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_encoder_tail_pack_f32x6_many': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr]),
    'sst_encoder_tail_bwd_workspace_bytes': (c_i64, [c_i64]),
    'sst_encoder_tail_fwd_f32x6': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_tail_bwd_f32x6': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_tail_pack_bf16_bytes': (c_i64, []),
    'sst_encoder_tail_pack_bf16': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_encoder_tail_pack_bf16_many': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr]),
    'sst_encoder_tail_bwd_bf16_workspace_bytes': (c_i64, [c_i64]),
    'sst_encoder_tail_fwd_bf16': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_tail_bwd_bf16': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_layer_bwd_bf16_workspace_bytes': (c_i64, [c_i64]),
    'sst_encoder_layer_fwd_bf16': (c_i32, [c_ptr, c_ptr]),
    'sst_encoder_layer_bwd_bf16': (c_i32, [c_ptr, c_ptr]),
    'sst_segment_reduce_work_words': (c_i64, [c_i64, c_i64, c_i32]),
    'sst_segment_reduce_fwd_work_f32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr,
                                                c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    'sst_vfe_linear_moments_f32': (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_bn_prepare_from_partials_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr,
                                                 c_ptr]),
    'sst_bn_stats_from_partials_f32': (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_bn_act_pool_bwd_reduce_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i32,
                                               c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_bn_act_pool_bwd_apply_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                              c_ptr, c_f32, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'sst_tall_linear_add_rows_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_ptr,
                                               c_i64, c_ptr]),
    'sst_segment_long_scratch_bytes': (c_i64, [c_i64, c_i64, c_i32]),
    'sst_segment_reduce_long_f32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr,
                                            c_ptr]),
    'sst_segment_reduce_profile_next': (c_i32, [c_ptr, c_ptr]),
    'sst_segment_reduce_bwd_f32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i32,
                                           c_ptr, c_ptr, c_ptr]),
    'sst_ingroup_rank_workspace_bytes': (c_i64, [c_i64]),
    'sst_ingroup_rank_i64': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr]),
    'sst_window_coors': (c_i32, [c_ptr, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_window_order_i32': (c_i32, [c_ptr, c_i32, c_i32, c_ptr, c_ptr]),
    'sst_region_batching_workspace_bytes': (c_i64, [c_i64]),
    'sst_region_batching': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i32] + [c_ptr] * 15 + [c_ptr, c_ptr]),
    'sst_frame_windows_per_sample': (c_i64, [c_ptr, c_ptr]),
    'sst_frame_voxels_i32': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr,
                                     c_ptr, c_ptr, c_ptr]),
    'sst_window_plan_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'sst_window_plan_i32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_i32, ctypes.c_uint32] + [c_ptr] * 11),
    'sst_sra_attn_fwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i32,
                                     c_f32, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_bwd_f32': (c_i32, [c_ptr] * 6 + [c_i64] * 5 + [c_ptr, c_ptr, c_i64, c_i64, c_i32, c_f32,
                                                                 c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                                                 c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_fwd_ord_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i32,
                                         c_f32, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_bwd_ord_f32': (c_i32, [c_ptr] * 6 + [c_i64] * 5 + [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_f32,
                                                                     c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                                                     c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_bwd_workspace_bytes': (c_i64, [c_i64, c_i32]),
    'sst_sra_attn_cos_fwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i32,
                                         c_ptr, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_cos_bwd_f32': (c_i32, [c_ptr] * 6 + [c_i64] * 5 + [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_ptr,
                                                                     c_i32, c_ptr, c_ptr, c_ptr,
                                                                     c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_fwd_bf16': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_f32, c_i32,
                                      c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_bwd_bf16': (c_i32, [c_ptr] * 6 + [c_i64] * 5 + [c_ptr, c_ptr, c_i64, c_i32, c_f32, c_i32, c_ptr, c_ptr,
                                                                  c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    'sst_sra_attn_bf16_profile_next': (c_i32, [c_i32, c_ptr, c_ptr]),
    'sst_sra_attn_fwd_ord_bf16': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_f32,
                                          c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_bwd_ord_bf16': (c_i32, [c_ptr] * 6 + [c_i64] * 5 + [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_f32, c_i32, c_ptr,
                                                                      c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    'sst_sra_attn_cos_fwd_bf16': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr,
                                          c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_sra_attn_cos_bwd_bf16': (c_i32, [c_ptr] * 6 + [c_i64] * 5 + [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_ptr,
                                                                      c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sst_add_layernorm_fwd_bf16': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr,
                                           c_ptr, c_ptr, c_ptr]),
    'sst_add_layernorm_bwd_bf16': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr,
                                           c_ptr]),
    'sst_cast_add_pos_bf16': (c_i32, [c_ptr, c_i32, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_weight_grad_group_workspace_bytes': (c_i64, [c_ptr, c_i32]),
    'sst_weight_grad_group_f32': (c_i32, [c_ptr, c_i32, c_ptr, c_ptr]),
    'sst_weight_grad_group_f32x6_workspace_bytes': (c_i64, [c_ptr, c_i32]),
    'sst_weight_grad_group_f32x6': (c_i32, [c_ptr, c_i32, c_ptr, c_ptr]),
    'sst_tall_linear_ln_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_f32, c_ptr,
                                       c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_tall_linear_ln_f32x3': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_f32, c_ptr,
                                         c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_tall_linear_epi_f32x3': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr,
                                          c_i64, c_ptr, c_i64, c_ptr]),
    'sst_tall_linear_ln_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_f32, c_ptr,
                                         c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_tall_linear_epi_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr,
                                          c_i64, c_ptr, c_i64, c_ptr]),
    'sst_tall_linear_epi2_f32x6': (c_i32, [c_ptr, c_ptr, c_i32, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_i32, c_i32, c_i32,
                                           c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'sst_add_layernorm_bwd2_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_tall_linear_epi_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr,
                                        c_i64, c_ptr, c_i64, c_ptr]),
    'sst_tall_linear_bf16': (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_i64, c_ptr,
                                     c_i64, c_ptr]),
    'sst_tall_linear_ln_bf16': (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_ptr,
                                        c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_wgrad_group_workspace_bytes': (c_i64, [c_ptr, c_i32]),
    'sst_wgrad_group_bf16': (c_i32, [c_ptr, c_i32, c_ptr, c_ptr]),
    'sst_cast_group_bf16': (c_i32, [c_ptr, c_i32, c_ptr]),
    'sst_gather_rows_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_f32, c_ptr, c_i64, c_ptr]),
    'sst_add_table_rows_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr]),
    'sst_scatter_rows_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr]),
    'sst_add_layernorm_fwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_f32, c_ptr, c_ptr, c_ptr,
                                          c_ptr]),
    'sst_add_layernorm_bwd_workspace_bytes': (c_i64, [c_i64, c_i32]),
    'sst_add_layernorm_pos_fwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                              c_ptr, c_ptr]),
    'sst_add_layernorm_act_fwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_f32, c_i32, c_ptr, c_ptr, c_ptr,
                                              c_ptr]),
    'sst_add_layernorm_act_bwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i64, c_i32, c_ptr, c_ptr, c_ptr,
                                              c_ptr, c_ptr]),
    'sst_add_layernorm_bwd_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr,
                                          c_ptr]),
    'sst_weight_grad_workspace_bytes': (c_i64, [c_i64, c_i32, c_i32]),
    'sst_weight_grad_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_colsum_workspace_bytes': (c_i64, [c_i64, c_i32]),
    'sst_colsum_f32': (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, c_ptr]),
    'sst_tall_linear_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i64,
                                    c_ptr]),
    'sst_tall_linear_gelu_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr,
                                         c_i64, c_ptr]),
    'sst_connected_components_workspace_bytes': (c_i64, [c_i64]),
    'sst_connected_components_xy_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_dynamic_point_pool_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'sst_dynamic_point_pool_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i32, c_i64, c_ptr,
                                           c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_candidates_i32': (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr]),
    'sst_spconv_inverse_to_map_i32': (c_i32, [c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_spconv_grid_subm_i32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_grid_conv_workspace_bytes': (c_i64, [c_i64]),
    'sst_spconv_grid_conv_count_i32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr,
                                               c_ptr, c_ptr]),
    'sst_spconv_grid_conv_maps_i32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr,
                                              c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_subm_map_i32': (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_invert_map_i32': (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_ptr, c_ptr]),
    'sst_spconv_pair_lists_workspace_bytes': (c_i64, [c_i32, c_i64]),
    'sst_spconv_pair_lists_i32': (c_i32, [c_ptr, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_gather_gemm_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr,
                                           c_i64, c_i32, c_ptr]),
    'sst_spconv_conv_os_workspace_bytes': (c_i64, [c_i32, c_i32, c_i32]),
    'sst_spconv_conv_os_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_i64,
                                       c_i32, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_conv_os_tile_rows': (c_i32, [c_i64, c_i32, c_i32]),
    'sst_spconv_conv_os_f32x3': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_i64,
                                         c_i32, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_conv_os_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_i64,
                                         c_i32, c_ptr, c_ptr, c_ptr]),
    'sst_spconv_conv_os_rows_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_i64,
                                              c_i32, c_ptr, c_ptr, c_i64, c_ptr]),
    'sst_spconv_conv_os_f32x6_workspace_bytes_rows': (c_i64, [c_i32, c_i32, c_i32, c_i64]),
    'sst_spconv_conv_os_f32x6_workspace_bytes': (c_i64, [c_i32, c_i32, c_i32]),
    'sst_spconv_os_tile_work_i32': (c_i32, [c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr]),
    'sst_spconv_maxpool_fwd_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    'sst_spconv_maxpool_bwd_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr,
                                           c_i64, c_ptr]),
    'sst_spconv_wgrad_os_workspace_bytes': (c_i64, [c_i32, c_i64, c_i64, c_i32, c_i32]),
    'sst_spconv_wgrad_os_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32,
                                        c_ptr, c_ptr, c_ptr]),
    'sst_spconv_wgrad_os_f32x6': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32,
                                        c_ptr, c_ptr, c_ptr]),
    'sst_spconv_wgrad_workspace_bytes': (c_i64, [c_i32, c_i64, c_i64, c_i32, c_i32]),
    'sst_spconv_wgrad_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i32, c_i32, c_i32,
                                     c_ptr, c_ptr, c_ptr]),
    'sst_concat_gather_f32': (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    'sst_recover_bev_f32': (c_i32, [c_ptr, c_i64, c_ptr, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                    c_ptr]),
    'sst_recover_bev_bwd_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    'sst_vfe_decorate_f32': (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_ptr, c_i64, ctypes.c_float, c_ptr, c_i32, c_i64,
                                     c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    'sst_event_create': (c_ptr, []),
    'sst_event_destroy': (None, [c_ptr]),
    'sst_event_elapsed_ms': (ctypes.c_float, [c_ptr, c_ptr]),
    'sst_sra_attn_profile_next_fwd': (c_i32, [c_ptr, c_ptr]),
    'sst_sra_attn_profile_next_bwd': (c_i32, [c_ptr, c_ptr]),
    'sst_bn_workspace_bytes': (c_i64, [c_i64, c_i32]),
    'sst_bn_stats_f32': (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_bn_act_fwd_f32': (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, c_i32, c_ptr, c_i64, c_ptr]),
    'sst_bn_act_bwd_reduce_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                          c_i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_bn_act_bwd_apply_f32': (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                         c_ptr, c_ptr, ctypes.c_float, c_i32, c_ptr, c_i64, c_ptr]),
    'sst_bn_prepare_f32': (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, ctypes.c_float, c_ptr, c_ptr,
                                   ctypes.c_float, c_ptr, c_ptr, c_ptr]),
    'sst_bn_prepare_tracked_f32': (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, ctypes.c_float, c_ptr, c_ptr,
                                           ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_bn_act_res_fwd_f32': (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i32, c_ptr, c_i64,
                                       c_ptr]),
    'sst_bn_act_res_bwd_reduce_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_i64, c_ptr, c_ptr,
                                              c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sst_bn_act_res_bwd_apply_f32': (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_i64, c_ptr, c_ptr,
                                             c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_float, c_i32, c_ptr, c_i64, c_ptr,
                                             c_i64, c_ptr]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def build(force=False, verbose=False):
    """Compile libsst_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(['make', '-C', _CSRC, 'clean'], check=True, capture_output=not verbose)
    r = subprocess.run(['make', '-C', _CSRC, '-j8'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('building libsst_amd.so failed:\n' + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('SST_AMD_LIB', LIB_PATH)  # developer override: A/B builds of the same library
    if not os.path.exists(path):
        raise RuntimeError(
            f'{path} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C sst_amd/csrc`). sst_amd has no CPU / eager fallback.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def version():
    return load().sst_version().decode()


class SSTError(RuntimeError):
    pass


SST_ERR_ARG, SST_ERR_UNSUPPORTED, SST_ERR_KEYSPACE = -1, -2, -3
_ERR = {-1: 'SST_ERR_ARG (invalid argument)', -2: 'SST_ERR_UNSUPPORTED', -3: 'SST_ERR_KEYSPACE'}


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f'hipError_t {rc}' if rc > 0 else f'error {rc}')
        raise SSTError(f'{what} failed: {msg}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """torch's current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Python Stream
    object per call (~9 us: 0.8 ms per fp32 step, 2.6 ms per bf16 step of pure launch-path overhead - the reduced-precision leg
    was host-bound on slower hosts); the raw accessor answers in ~0.3 us."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


class Fp32Master(object):
    """Mix-in of the entry modules (voxel encoders, input layers, backbones): what ``model.half()`` means for them.

    The reference's half-precision training (``fp16 = dict(loss_scale=32.0)``, configs/sst_refactor/
    sst_waymoD5_1x_3class_8heads_v2.py:82) goes through mmcv's ``wrap_fp16_model`` - ``model.half()`` plus
    ``fp16_enabled = True`` on every module that has the attribute - and ``Fp16OptimizerHook``, which keeps fp32 master copies
    of the weights and scales the loss.  On this hardware the encoder layers' reduced-precision mode is bf16 storage with fp32
    accumulation (sst_amd/bf16.py) and the parameters of these modules ARE the fp32 masters: a conversion of the module to
    float16 is therefore remembered (``half_requested``) and NOT applied - parameters and buffers stay fp32, the encoder stack
    switches to its bf16 mode where the reference's ``auto_fp16`` (sst_basic_block_v2.py:102-104) would cast to half, and
    outputs are handed on as float16 where the reference's layers would (the modules behind - neck, heads - are half).  The
    hook's bookkeeping (copy_grads_to_fp32 / copy_params_to_fp16) works unchanged on fp32 parameters.  bf16 has fp32's
    exponent range: the loss scale is harmless (and not needed)."""

    half_requested = False

    def _apply(self, fn, *args, **kwargs):
        try:
            to_half = fn(torch.zeros(1, dtype=torch.float32)).dtype == torch.float16
        except Exception:
            to_half = False
        if to_half:
            for m in self.modules():
                if isinstance(m, Fp32Master):
                    m.half_requested = True
            return self
        return super()._apply(fn, *args, **kwargs)


def wants_half(module):
    """did the caller ask for the reference's fp16 mode: ``model.half()`` (remembered by Fp32Master) or ``fp16_enabled`` set on
    one of the module's encoder layers / the module itself by mmcv's wrap_fp16_model"""
    if getattr(module, 'half_requested', False) or getattr(module, 'fp16_enabled', False):
        return True
    # the submodules that carry the flag at all (the reference's auto_fp16 modules: encoder layers, norms, the input layer): found
    # once per module object - walking module.modules() on every forward call was 7 x ~150 modules per step, 0.4 ms of a 3.4 ms
    # host step (tools/host_profile.py).  A module added later is seen after `del module.__dict__['_sst_fp16_watch']`.
    watch = module.__dict__.get('_sst_fp16_watch')
    if watch is None:
        watch = module.__dict__['_sst_fp16_watch'] = [m for m in module.modules() if hasattr(m, 'fp16_enabled')]
    for m in watch:
        if m.fp16_enabled:
            return True
    return False


def as_fp32(*tensors):
    """the reference's force_fp32 on inputs (voxel_encoder.py:229, dynamic_voxelnet.py:50): half tensors are cast, others pass"""
    out = tuple(t.float() if (t is not None and torch.is_tensor(t) and t.dtype in (torch.float16, torch.bfloat16)) else t
                for t in tensors)
    return out if len(out) != 1 else out[0]


def refuse_fp16(module=None, *tensors, what=None):
    """kept for callers outside the entry modules: half PARAMETERS cannot occur in an Fp32Master module; a half tensor handed to
    a kernel wrapper directly is a usage error there (the entry modules cast with as_fp32 instead)"""
    for t in tensors:
        if t is not None and torch.is_tensor(t) and t.dtype == torch.float16:
            raise RuntimeError(what or 'sst_amd: float16 tensor handed to an fp32 kernel wrapper - cast it (the entry modules do: '
                                       'sst_amd._lib.as_fp32), or use the bf16 mode of the encoder stack (set_precision)')


def require_cuda(*tensors):
    """CHECK_INPUT of the reference (scatter_points_cuda.cu:9-15): device + contiguous, else RuntimeError."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('sst_amd: tensor must be a CUDA/HIP tensor (no CPU path in this library)')
        if not t.is_contiguous():
            raise RuntimeError('sst_amd: tensor must be contiguous')


def farray(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def i32array(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def i64array(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
