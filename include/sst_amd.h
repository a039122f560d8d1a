/*
 * sst_amd.h — C ABI of libsst_amd.so: the MI355X (gfx950) implementation of the SST / FSD sparse
 * hot path (dynamic voxelization -> point->voxel scatter -> window bucketing -> Sparse Regional
 * Attention core -> SIR segmented max).
 *
 * Conventions
 *   - Every pointer named d_* is a DEVICE pointer (HBM). h_* pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream). All work is
 *     enqueued on that stream; no entry point synchronises the device unless its comment says so.
 *   - All buffers are owned by the caller (reference contract: tensors are caller/allocator owned,
 *     SURVEY.md §8b "Ownership"). Workspaces are sized with the matching *_workspace_bytes().
 *   - Return value: 0 on success, a positive hipError_t value if the HIP runtime reported an error,
 *     or a negative SST_ERR_* code for an argument the library refuses.
 *   - Row-major, contiguous unless a stride argument (in elements) is given.
 *
 * Each entry point cites the reference interface (file:line under the reference tree) it replaces.
 */
#ifndef SST_AMD_H
#define SST_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SST_OK 0
#define SST_ERR_ARG (-1)          /* invalid argument (null pointer, negative size, bad mode)        */
#define SST_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels are built for                    */
#define SST_ERR_KEYSPACE (-3)     /* coordinates do not pack into 63 bits                             */

typedef enum { SST_REDUCE_SUM = 0, SST_REDUCE_MEAN = 1, SST_REDUCE_MAX = 2 } sst_reduce_t;

/* Library self-description: returns a static string "sst_amd <version> gfx950". */
const char* sst_version(void);

/* ------------------------------------------------------------------------------------------------
 * (a1) dynamic voxelization.
 * Replaces voxel_layer.dynamic_voxelize (mmdet3d/ops/voxel/src/voxelization.h:71-83 ->
 * dynamic_voxelize_gpu, voxelization_cuda.cu:332-375, kernel :24-65).
 *   c[d] = clamp((int)floorf((p[d] - range[d]) / voxel_size[d]), 0, grid[d]-1), fp32 subtract then
 *   fp32 IEEE divide; grid[d] = ceil((range[d+3]-range[d]) / voxel_size[d]) in fp32 (cuda.cu:355-357);
 *   THIS FORK clamps out-of-range points into the border voxel (cuda.cu:37-59).
 * d_points: [n, row_stride] fp32, columns 0..2 = x,y,z.  d_coors: [n, coors_stride] int32; columns
 * coors_col0..coors_col0+2 receive (z, y, x).  If batch_idx >= 0 and coors_col0 == 1, column 0 is
 * filled with batch_idx (fuses DynamicVoxelNet.voxelize's F.pad, detectors/dynamic_voxelnet.py:49-71).
 * ---------------------------------------------------------------------------------------------- */
int sst_dynamic_voxelize_f32(const float* d_points, int64_t n, int64_t row_stride,
                             const float voxel_size[3], const float coors_range[6],
                             int32_t* d_coors, int64_t coors_stride, int coors_col0, int batch_idx,
                             void* stream);
/* Host helper: the grid the kernel clamps to (fp32 ceil, as voxelization_cuda.cu:355-357). */
void sst_dynamic_voxelize_grid(const float voxel_size[3], const float coors_range[6], int32_t grid_xyz[3]);

/* ------------------------------------------------------------------------------------------------
 * Device primitives shared by the unique / bucketing steps (exported so they can be tested alone).
 * ---------------------------------------------------------------------------------------------- */
/* Exclusive prefix sum of int32.  d_out may alias d_in.  d_total (optional, device int32) receives
 * the grand total.  Workspace: sst_scan_workspace_bytes(n). */
int64_t sst_scan_workspace_bytes(int64_t n);
int sst_exclusive_scan_i32(const int32_t* d_in, int32_t* d_out, int64_t n, int32_t* d_total,
                           void* d_workspace, void* stream);

/* Stable LSD radix sort of (key, index) pairs on the low `key_bits` bits of 64-bit keys.
 * On return d_keys_out is sorted ascending and d_perm_out[i] is the original position of the i-th
 * smallest key (ties in ascending original position).  d_keys_in is clobbered (used as ping-pong).
 * Workspace: sst_sort_workspace_bytes(n). */
int64_t sst_sort_workspace_bytes(int64_t n);
int sst_sort_pairs_u64(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_perm_out, int64_t n,
                       int key_bits, void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a3/a5) sorted-unique of integer coordinate rows — the part of
 *   dynamic_point_to_voxel_forward_gpu that calls at::unique_dim (scatter_points_cuda.cu:202-205) and of
 *   scatter_v2 that calls torch.unique(dim=0) (ops/sst/sst_ops.py:151-165).
 * Rows are packed into one 64-bit key  key = 1 + sum_j (c_j - mins[j]) * stride_j  (last column fastest),
 * so ascending key order == lexicographic row order.  With invalid_if_negative != 0, any row with a
 * negative entry gets key 0 (the reference's coors.masked_fill(any(<0), -1): all such rows collapse
 * into ONE group that sorts first, scatter_points_cuda.cu:200).
 * invalid_if_negative == 2 is the batched form (DynamicScatter.forward's per-sample loop,
 * ops/voxel/scatter_points.py:85-99, done in one pass): column 0 is the batch index, the validity test
 * looks at columns 1.., and an invalid row becomes (b,-1,-1,...) — pass mins[j>=1] = -1 so that it sorts
 * first inside its own sample.
 * coor_is_i64: 0 -> int32 rows, 1 -> int64 rows.   extents[j] = (max_j - mins[j] + 1) upper bound.
 * Outputs (all device, caller allocated):
 *   d_perm   [n]   uint32  row indices grouped by unique row, ascending row index inside a group
 *   d_inverse[n]   int32   group id of every input row (== torch.unique's inverse)
 *   d_offsets[n+1] int32   CSR offsets into d_perm; entries 0..M valid
 *   d_ukeys  [n]   uint64  packed key of each group; entries 0..M-1 valid
 *   d_num_unique   int32   M
 * Workspace: sst_unique_workspace_bytes(n).  No host synchronisation.
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_unique_workspace_bytes(int64_t n);
int sst_unique_rows(const void* d_coors, int coor_is_i64, int64_t n, int ncols, int64_t row_stride,
                    const int64_t* h_mins, const int64_t* h_extents, int invalid_if_negative,
                    uint32_t* d_perm, int32_t* d_inverse, int32_t* d_offsets, uint64_t* d_ukeys,
                    int32_t* d_num_unique, void* d_workspace, void* stream);
/* Decode packed keys back into rows (int32 or int64 output); key 0 decodes to all -1. */
int sst_unpack_keys(const uint64_t* d_ukeys, int64_t m, int ncols, const int64_t* h_mins,
                    const int64_t* h_extents, void* d_rows_out, int out_is_i64, int64_t out_stride,
                    int out_col0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a3/a5/a15) segmented reduce over the CSR produced above: one output row per group.
 * Replaces feats_reduce_kernel (scatter_points_cuda.cu:80-103, float-CAS atomics) and
 * torch_scatter.scatter_max / scatter(reduce=sum|mean) (call sites ops/sst/sst_ops.py:172-177).
 *   d_feats [n, c] fp32; group g reduces rows d_perm[d_offsets[g] .. d_offsets[g+1]).
 *   MAX of an empty group is -inf, SUM/MEAN is 0.  MEAN = sum / (float)count (cuda.cu:228-229).
 *   d_argmax (optional, [m, c] int32): row index of the FIRST (smallest index) row attaining the max
 *   (the tie rule of max_reduce_traceback_scatter_idx_kernel, cuda.cu:135-160); n for empty groups.
 *   d_group_index (optional, [m] int32): output row g reduces CSR group d_group_index[g] instead of group g (a negative
 *   entry: an empty group)
 *   (a subset / re-ordering of the groups without gathering the result: the batched DynamicScatter keeps all but
 *   the first sorted-unique row of every sample).
 * ---------------------------------------------------------------------------------------------- */
int sst_segment_reduce_fwd_f32(const float* d_feats, int64_t n, int c, const uint32_t* d_perm,
                               const int32_t* d_offsets, const int32_t* d_group_index, int64_t m, int mode,
                               float* d_out, int32_t* d_argmax, const int32_t* d_m_limit, void* stream);
/* sst_segment_reduce_fwd_work_f32: sst_segment_reduce_fwd_f32 made robust against a few very long groups among many short
 * ones (the voxels next to the sensor of a real LiDAR sweep hold thousands of points while the average voxel holds 1-10:
 * voxel_encoder.py:185-298 reduces them with the same DynamicScatter).  Groups of more than 32 rows are put on the work
 * list d_work by the element kernel - cut into chunks of 512 rows - and reduced by one workgroup per chunk in a second launch
 * of fixed size; a third merges the chunks of the groups that were cut, in chunk order; same tie rule, same results up to the
 * association of a cut group's sum.  d_work: int32 [work_capacity], 16-byte aligned, work_capacity >=
 * sst_segment_reduce_work_words(n, m, c), its first 8 entries ZEROED ONCE by the caller; the kernels leave them zeroed (reusable
 * by every later call on the same stream).  d_work == NULL: exactly
 * sst_segment_reduce_fwd_f32.  d_scale / d_shift (optional, [c], c % 4 == 0, with a work list): every value is read as
 * relu(x * scale + shift) - the BatchNorm + ReLU of DynamicVFE's last layer applied while its output is pooled
 * (voxel_encoder.py:286-296), whose activated [n, c] matrix is then never written. */
int64_t sst_segment_reduce_work_words(int64_t n, int64_t m, int c);
int sst_segment_reduce_fwd_work_f32(const float* d_feats, int64_t n, int c, const uint32_t* d_perm,
                                    const int32_t* d_offsets, const int32_t* d_group_index, int64_t m, int mode,
                                    float* d_out, int32_t* d_argmax, const int32_t* d_m_limit, int32_t* d_work,
                                    int64_t work_capacity, const float* d_scale, const float* d_shift, void* stream);
/* sst_segment_reduce_long_f32: the same reduction (all three modes, the arg-max tie rule) for groupings whose groups are
 * long and uneven (FSD's clusters, voxel_encoder.py:696-764 / backbones/sir.py:67-88: torch_scatter.scatter_max / scatter over
 * cluster ids): work is cut into tiles of sorted positions whatever group they belong to, groups inside a tile are written
 * directly, a group that crosses tiles is merged by the last of its tiles to arrive (one atomic ticket per crossing group and
 * tile; deterministic merge order; csrc/scatter.hip).  d_inverse [n] int32 = group of every point (+ inverse_shift; negative =
 * not reduced); every group with at least one point is written, groups without points are NOT.  d_scratch:
 * sst_segment_long_scratch_bytes(n, m, c) bytes, 256-byte aligned, its first 4 m bytes ZEROED ONCE by the caller; the kernel
 * leaves them zeroed (reusable by every later call on the same grouping without a fill).  c <= 256. */
int64_t sst_segment_long_scratch_bytes(int64_t n, int64_t m, int c);
int sst_segment_reduce_long_f32(const float* d_feats, int64_t n, int c, const uint32_t* d_perm, const int32_t* d_inverse,
                                int inverse_shift, const int32_t* d_offsets, int64_t m, int mode, void* d_scratch,
                                float* d_out, int32_t* d_argmax, void* stream);
/* sst_segment_reduce_profile_next(start, stop): one-shot hipEvent_t pair bound to the NEXT forward launch of this thread
 * (kernel start / stop; measurement hook of bench.py's FSD workloads, as sst_sra_attn_profile_next_fwd). */
int sst_segment_reduce_profile_next(void* start, void* stop);
/* Backward (scatter_points_cuda.cu:236-303).  d_grad_feats [n, c] is fully written (zero where no
 * gradient flows).  SUM/MEAN: g[i] = G[inv[i]] (/count); rows with d_inverse[i] < 0 get 0.
 * MAX: gradient goes to d_argmax[g, ch] only.  d_inverse may be shifted by the caller
 * (inverse_shift is added before use; the DynamicScatter "first row" quirk uses -1).  With d_group_index the
 * inverse map must already address OUTPUT rows; the group index is only used for the MEAN count.
 * d_m_limit (both directions, may be NULL): the number of output rows when it is only known on the device - `m`
 * is then an upper bound that sizes the launch and rows >= *d_m_limit are neither computed nor read. */
int sst_segment_reduce_bwd_f32(const float* d_grad_out, int64_t m, int c, const int32_t* d_inverse,
                               int inverse_shift, const int32_t* d_offsets, const int32_t* d_group_index,
                               const int32_t* d_argmax, int64_t n, int mode, float* d_grad_feats,
                               const int32_t* d_m_limit, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a15 / f1) d_out[i] = [ d_x[i, 0:c1] | d_g[d_idx[i], 0:c2] ]  ([n, c1 + c2] contiguous): the gather by the inverse map
 * + concatenation between the layers of DynamicVFE / DynamicScatterVFE / SIRLayer (voxel_encoders/voxel_encoder.py:
 * 288-291, 605-607, 745-750) in one coalesced pass.  d_idx[i] < 0 reads row 0.  c1, c2 multiples of 4, 16-byte rows.
 * ---------------------------------------------------------------------------------------------- */
int sst_concat_gather_f32(const float* d_x, int64_t ldx, int c1, const float* d_g, int64_t ldg, int c2,
                          const int32_t* d_idx, int64_t n, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a14) SSTv2.recover_bev (models/backbones/sst_v2.py:161-197): voxel features [m, c] + coordinates (b, z, y, x)
 * -> dense canvas of batch x ny x nx cells, written ONCE, cell by cell (the voxel's row or zeros), in CHANNELS-LAST
 * order: d_canvas is [batch, ny, nx, c] contiguous = the channels-last memory of the reference's logical
 * [batch, c, ny, nx] tensor (the host layer hands it on as that view).  c % 4 == 0, 16-byte aligned rows.
 *   d_cell_map [batch*ny*nx] int32 scratch (cell -> voxel, -1 empty); d_cell_of_voxel [m] int32 out: the cell each
 *   voxel was written to (-1: outside the canvas, or another voxel with the same cell was written instead - the
 *   reference's index assignment keeps one of them too).  Backward: rows of the canvas gradient at those cells.
 * ---------------------------------------------------------------------------------------------- */
int sst_recover_bev_f32(const float* d_feats, int64_t ldf, const void* d_coors, int coor_is_i64, int64_t ldc, int64_t m,
                        int batch, int ny, int nx, int c, int32_t* d_cell_map, int32_t* d_cell_of_voxel, float* d_canvas,
                        void* stream);
int sst_recover_bev_bwd_f32(const float* d_grad_canvas, const int32_t* d_cell_of_voxel, int64_t m, int c,
                            float* d_grad_feats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (f1) Decorate step of DynamicVFE / DynamicScatterVFE (voxel_encoders/voxel_encoder.py:252-271, :569-589) in one
 * launch: d_out[i] = [ point features (c) | xyz - mean xyz of the point's voxel, divided by cluster_div (3, if
 * with_cluster) | xyz - centre of the point's voxel (3, if with_center) ], bit-identical to the composed torch ops.
 *   d_inverse [n] int32 voxel of every point (< 0 reads voxel 0, like the reference's zero-initialised canvas);
 *   d_voxel_mean [m, >= 3] (row stride ldm); d_coors [n, 4] (b, z, y, x) int32 or int64 (row stride ldc);
 *   voxel_size / offsets: HOST float[3] (vx, vy, vz) and (v / 2 + range_min) per axis.
 * ---------------------------------------------------------------------------------------------- */
int sst_vfe_decorate_f32(const float* d_points, int64_t ldp, int64_t n, int c, const int32_t* d_inverse,
                         const float* d_voxel_mean, int64_t ldm, float cluster_div, const void* d_coors,
                         int coor_is_i64, int64_t ldc, const float* voxel_size, const float* offsets,
                         int with_cluster, int with_center, float* d_out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a7) in-group rank.  Replaces TorchEx ingroup_indices.forward(group_inds, out_inds)
 * (call site ops/sst/sst_ops.py:244-264).  d_rank[i] = number of j < i with group[j] == group[i]
 * (the stable choice; the reference leaves the order unspecified).  group ids must lie in [0, 2^key_bits).
 * Workspace: sst_ingroup_rank_workspace_bytes(n).
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_ingroup_rank_workspace_bytes(int64_t n);
int sst_ingroup_rank_i64(const int64_t* d_group, int64_t n, int key_bits, int64_t* d_rank,
                         void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a6) window coordinates, both shifts in one launch.  Replaces get_window_coors
 * (ops/sst/sst_ops.py:266-314) called twice by SSTInputLayerV2.window_partition
 * (middle_encoders/sst_input_layer_v2.py:229-236).
 *   d_coors [m, 4] (b,z,y,x) int32 or int64.  sparse_shape = (sx,sy,sz), window_shape = (wx,wy,wz).
 *   d_win[s]   [m] int32     batch_win_inds for shift s (0: no shift, 1: half-window shift)
 *   d_ciw[s]   [m,3] int32   coors_in_win (z,y,x)
 * ---------------------------------------------------------------------------------------------- */
int sst_window_coors(const void* d_coors, int coor_is_i64, int64_t m, const int32_t sparse_shape[3],
                     const int32_t window_shape[3], int32_t* d_win0, int32_t* d_ciw0,
                     int32_t* d_win1, int32_t* d_ciw1, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a7-a9) region batching for both shifts: drop levels, voxel drop, in-window order, and the
 * window CSR ("plan") the SRA kernels consume.  Replaces SSTInputLayerV2.drop_voxel /
 * drop_single_shift (sst_input_layer_v2.py:128-226) and get_flat2win_inds (sst_ops.py:26-64).
 *   levels: n_levels rows of (max_tokens, lower, upper): level l applies when lower <= count < upper.
 *   Inputs d_win0/d_win1 [m] from sst_window_coors; win ids must be < 2^win_bits.
 * Outputs (device, caller allocated, sizes in elements):
 *   d_keep     [m] int32   1 if the voxel survives both shifts
 *   d_newidx   [m] int32   index among survivors (exclusive scan of keep)
 *   d_level[s] [m] int32   drop level of the voxel's window in shift s (-1 if no range matches)
 *   d_inner[s] [m] int32   rank among SURVIVORS of its window (valid where keep)
 *   d_flat2win[s] [m] int32  contiguous-window-id-within-level * max_tokens + inner (valid where keep)
 *   d_tok[s]   [m] int32   survivor indices (NEW numbering) grouped by window, ascending window id,
 *                           ascending inner; first M' entries valid
 *   d_winoff[s][m+1] int32 CSR offsets into d_tok[s] per non-empty window; first W_s+1 entries valid
 *   d_winlevel[s][m] int32 level of each non-empty window; first W_s valid
 *   d_counts   [8] int32   {M', W_0, W_1, T_0, T_1, 0...}: survivors, windows per shift, largest surviving window
 *                            population per shift (lets the caller pick the attention kernels' tile class)
 * Semantics follow the reference exactly: shift-1 levels are computed on the survivors of shift 0,
 * shift-0 levels are NOT recomputed after the second filter (sst_input_layer_v2.py:186-194).
 * Workspace: sst_region_batching_workspace_bytes(m).
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_region_batching_workspace_bytes(int64_t m);
int sst_region_batching(const int32_t* d_win0, const int32_t* d_win1, int64_t m, int win_bits,
                        const int32_t* h_levels /*[n_levels,3]*/, int n_levels,
                        int32_t* d_keep, int32_t* d_newidx,
                        int32_t* d_level0, int32_t* d_level1, int32_t* d_inner0, int32_t* d_inner1,
                        int32_t* d_flat2win0, int32_t* d_flat2win1,
                        int32_t* d_tok0, int32_t* d_tok1, int32_t* d_winoff0, int32_t* d_winoff1,
                        int32_t* d_winlevel0, int32_t* d_winlevel1, int32_t* d_counts,
                        void* d_workspace, void* stream);
/* Launch order of the windows of a CSR for the register-resident attention kernels: d_order [n_windows] = window ids by ascending
 * token count (winoff[w + 1] - winoff[w] <= max_tokens), ties in id order - torch.sort(sizes, stable=True)[1] of the host layer
 * (the kernels dispatch from the end: largest windows first).  One launch.  SST_ERR_UNSUPPORTED: max_tokens >= 512. */
int sst_window_order_i32(const int32_t* d_winoff, int n_windows, int max_tokens, int32_t* d_order, void* stream);


/* ------------------------------------------------------------------------------------------------
 * (a2-a9 in one piece) Index plan of a frame batch without host round trips (csrc/frame_plan.hip): the same voxel
 * table / window bucketing / voxel drop / window CSR as sst_unique_rows + sst_window_coors + sst_region_batching,
 * from buffers sized by host-known upper bounds; the counts stay on the device (d_counts) and are read once.
 * Replaces, for the SST training path, DynamicScatter's per-sample unique bookkeeping (ops/voxel/scatter_points.py:
 * 85-99, src/scatter_points_cuda.cu:202-210) and SSTInputLayerV2.window_partition / drop_voxel / get_flat2win_inds
 * (middle_encoders/sst_input_layer_v2.py:128-236, ops/sst/sst_ops.py:26-64, 266-331).
 *
 * sst_frame_voxels_i32: from the sorted-unique groups of the point coordinates (sst_unique_rows with
 *   invalid_if_negative = 2, mins (0,-1,-1,-1), extents (B, gz+1, gy+1, gx+1); d_ukeys / d_inverse = its sorted keys and its
 *   point -> group map, d_num_groups = its device count):
 *   drop_mode 1: the first group of every sample is discarded (the reference's unconditional out_coors[1:]),
 *   drop_mode 0: only the group of invalid rows.  Kept groups are the voxels v = 0..M-1 (ascending key):
 *     d_vcoors [n_points, 4] int32 (b,z,y,x), d_gidx [n_points] group of voxel v, d_coors_map [n_points] voxel of every
 *     point (-1: dropped), d_grid [B*gz*gy*gx] dense cell -> voxel map (-1: empty), d_counts[0] = M,
 *     d_dropped_groups (optional, [B]): the discarded group of every sample, -1 where a sample has none (its points read
 *     voxel row 0 through map_voxel_center_to_point's zero-initialised canvas, voxel_encoder.py:246-270, and hand their
 *     gradient to that row: sst_segment_reduce_fwd_f32 with this array as d_group_index gives that sum without atomics).
 * sst_window_plan_i32: window / shifted-window bucketing, drop levels (h_levels: n_levels x (max_tokens, lo, hi)), the
 *   three passes of drop_voxel, window CSR.  seed = 0: survivors of an over-full window are its first voxels in
 *   ascending voxel order; seed != 0: a uniformly random subset (the reference's shuffle_voxels).  Outputs, kept voxels
 *   numbered window-major (position in the shift-0 CSR):
 *     d_feat_index / d_feat_index32 [n_upper] voxel v of every output row, d_out_coors [n_upper, 4] int64,
 *     d_winoff0 / d_winoff1 [n_windows + 1] CSR offsets of the non-empty windows (shift-0 tokens are 0..M'-1 in order),
 *     d_tok1 [n_upper] output rows grouped by shift-1 window, d_posidx0/1 [n_upper] row of the positional table
 *     ((z_in * wy + y_in) * wx + x_in); d_counts: [1] M', [2] W_0, [3] W_1, [4] / [5] largest window of each shift.
 *   n_windows = B * sst_frame_windows_per_sample(grid, window).  Limits: B <= 64, window <= 512 cells,
 *   B * cells <= 2^28 (otherwise SST_ERR_UNSUPPORTED: callers use the piecewise entry points).
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_frame_windows_per_sample(const int32_t grid_zyx[3], const int32_t window_shape[3]);
int sst_frame_voxels_i32(const uint64_t* d_ukeys, const int32_t* d_inverse, const int32_t* d_num_groups, int64_t n_points, int batch_size, const int32_t grid_zyx[3],
                         int drop_mode, int32_t* d_vcoors, int32_t* d_gidx, int32_t* d_coors_map, int32_t* d_grid,
                         int32_t* d_counts, int32_t* d_dropped_groups, void* stream);
int64_t sst_window_plan_workspace_bytes(int64_t n_upper, int64_t n_windows);
int sst_window_plan_i32(const int32_t* d_vcoors, const int32_t* d_grid, int64_t n_upper, int batch_size,
                        const int32_t grid_zyx[3], const int32_t window_shape[3], const int32_t* h_levels, int n_levels,
                        uint32_t seed, int64_t* d_feat_index, int32_t* d_feat_index32, int64_t* d_out_coors,
                        int32_t* d_tok1, int32_t* d_winoff0, int32_t* d_winoff1, int32_t* d_posidx0, int32_t* d_posidx1,
                        int32_t* d_counts, void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a12) Sparse Regional Attention core over variable-length windows (no padding, no key mask):
 *   for every window w (tokens tok[winoff[w] .. winoff[w+1])), every head h:
 *       S = (Q_h * scale) K_h^T ; P = softmax_rows(S) ; O_h = P V_h
 * Replaces the per-drop-level  flat2window -> nn.MultiheadAttention bmm/softmax/bmm -> window2flat
 * of WindowAttention.forward (models/sst/sst_basic_block_v2.py:41-75); the in/out projections stay
 * GEMM-library calls in the host layer; the cosine variant (cosine_msa.py:159-170) is sst_sra_attn_cos_{fwd,bwd}_f32
 * below (normalisation and 1 / clamp(tau) inside the kernels); for layouts those do not take the host layer normalises and
 * rescales Q,K before calling this with scale = 1.  head_dim is 16 (d_model/nhead =
 * 128/8, 192/12 in every SST config); n_heads must be a multiple of 4.
 * Q,K,V,O: [M, n_heads*16] fp32 with row strides ldq/ldk/ldv/ldo (elements, multiples of 4; base
 * pointers 16-byte aligned).  d_lse [M, n_heads] fp32: log-sum-exp of each softmax row (for backward).
 *   max_tokens: upper bound on tokens per window the caller guarantees (0 = unknown).
 *   d_tok == NULL: the tokens of window w are the rows winoff[w] .. winoff[w+1] - 1 themselves (feature rows kept in
 *     window order, as sst_window_plan_i32 numbers them for the unshifted partition): no token list is read; taken by the
 *     register-resident kernels only (impl 0 / 3, 0 < max_tokens <= 144, aligned operands), else SST_ERR_UNSUPPORTED.
 *   impl: 0 = auto: register-resident MFMA kernels (wave = window x head, no LDS) for windows <= 144
 *             tokens, generic VALU kernel above that;
 *         1 = generic VALU kernel for every window (validation path);
 *         2 = LDS-staged MFMA kernels (K,V of a 4-head group in LDS; the first implementation);
 *         3 = forward as 0; backward as two register-resident launches (dQ, then dK / dV), kept for comparison.
 * ---------------------------------------------------------------------------------------------- */
int sst_sra_attn_fwd_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ldq, int64_t ldk,
                         int64_t ldv, const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows,
                         int n_heads, float scale, int max_tokens, int impl, float* d_o, int64_t ldo,
                         float* d_lse, void* stream);
/* Backward: given dO, recomputes P from (Q,K,LSE) and writes dQ, dK, dV for every token row listed in
 * d_tok (other rows untouched).  n_tokens = number of rows of the [M, *] tensors.  impl 0: ONE pass over
 * Q, K, V, O, dO per (window, head) wave (dQ, dK and dV from the same S / dP tiles; 8 x 64 B per token and
 * head of traffic); d_workspace may be NULL for it.
 * Workspace (impl 3): sst_sra_attn_bwd_workspace_bytes(n_tokens, n_heads) (rowsum(dO*O) per token, head). */
int64_t sst_sra_attn_bwd_workspace_bytes(int64_t n_tokens, int n_heads);
int sst_sra_attn_bwd_f32(const float* d_q, const float* d_k, const float* d_v, const float* d_o,
                         const float* d_do, const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv,
                         int64_t ldo, int64_t lddo, const int32_t* d_tok, const int32_t* d_winoff,
                         int64_t n_windows, int64_t n_tokens, int n_heads, float scale, int max_tokens,
                         int impl, float* d_dq, float* d_dk, float* d_dv, int64_t lddq, int64_t lddk,
                         int64_t lddv, void* d_workspace, void* stream);

/* The same two calls with a LAUNCH ORDER of the windows: d_win_order (may be NULL) is a permutation of 0 .. n_windows - 1;
 * the workgroups are dispatched from its END (so: windows sorted by ascending token count = the largest windows first,
 * which shortens the tail of the launch: -5 % on the bench frame).  Results do not depend on it.  Used by the
 * register-resident kernels (impl 0 / 3 forward, impl 0 backward); the other kernels ignore it. */
int sst_sra_attn_fwd_ord_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int n_heads, float scale, int max_tokens, int impl, float* d_o, int64_t ldo, float* d_lse,
                             void* stream);
int sst_sra_attn_bwd_ord_f32(const float* d_q, const float* d_k, const float* d_v, const float* d_o, const float* d_do,
                             const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int64_t n_tokens, int n_heads, float scale, int max_tokens, int impl, float* d_dq,
                             float* d_dk, float* d_dv, int64_t lddq, int64_t lddk, int64_t lddv, void* d_workspace,
                             void* stream);

/* Scaled cosine attention (CosineMultiheadAttention, models/sst/cosine_msa.py:123-185, 449-466; layer_cfg = dict(cosine=True,
 * tau_min=..) of configs/sst_refactor/sst_waymoD5_1x_3class_centerhead.py:75 and configs/fsd/fsd_waymoD1_1x_sst_encoder.py:70):
 *       S = normalize(Q_h) normalize(K_h)^T * head_scale[h] ; P = softmax_rows(S) ; O_h = P V_h
 * normalize = x / max(|x|_2, 1e-12) over the 16 channels of the head (torch.nn.functional.normalize), formed in the load
 * prologue of the register-resident kernels (one 4-lane reduction per row fragment).  d_head_scale: [n_heads] fp32 in DEVICE
 * memory = 1 / clamp(tau, tau_min) (a shared tau is passed expanded) - the parameter never visits the host.
 * Backward: d_dq / d_dk are the gradients of the UN-normalised q / k (d x = (d x^ - x^ (x^ . d x^)) / |x|, taken where the
 * row fragment is stored); d_r [n_tokens, n_heads] receives x^ . d x^ of the query side, whose column sum is the gradient of
 * the scale: d head_scale[h] = sum_rows d_r[:, h] / head_scale[h].  Same layouts and d_tok == NULL rule as above; windows of
 * <= 144 tokens, 16-byte aligned operands and n_heads % 4 == 0 only (else SST_ERR_UNSUPPORTED: the host layer then
 * normalises outside). */
int sst_sra_attn_cos_fwd_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int n_heads, const float* d_head_scale, int max_tokens, float* d_o, int64_t ldo, float* d_lse,
                             void* stream);
int sst_sra_attn_cos_bwd_f32(const float* d_q, const float* d_k, const float* d_v, const float* d_o, const float* d_do,
                             const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int64_t n_tokens, int n_heads, const float* d_head_scale, int max_tokens, float* d_dq, float* d_dk,
                             float* d_dv, int64_t lddq, int64_t lddk, int64_t lddv, float* d_r, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a12, reduced precision) The same attention core with bf16 storage: Q, K, V, O, dO, dQ, dK, dV are bf16 ([M, n_heads*16],
 * row strides in elements, multiples of 4; 8-byte aligned), the softmax, the log-sum-exp (fp32 [M, n_heads]) and every
 * accumulation are fp32; v_mfma_f32_16x16x16_bf16.  The reference's own training precision for these layers is fp16
 * (configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82).  Windows up to 144 tokens (max_tokens must say so, else
 * SST_ERR_UNSUPPORTED).  The backward is one pass per (window, head) wave (dQ, dK, dV from one read of Q, K, V, O, dO).
 * d_tok == NULL: the tokens of window w are the rows winoff[w] .. winoff[w+1] - 1 themselves (as for the fp32 kernels).
 * sst_sra_attn_bf16_profile_next(backward, start, stop): one-shot kernel-bound events, as for the fp32 kernels.
 * ---------------------------------------------------------------------------------------------- */
int sst_sra_attn_fwd_bf16(const void* d_q, const void* d_k, const void* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                          const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows, int n_heads, float scale,
                          int max_tokens, void* d_o, int64_t ldo, float* d_lse, void* stream);
int sst_sra_attn_bwd_bf16(const void* d_q, const void* d_k, const void* d_v, const void* d_o, const void* d_do,
                          const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                          const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows, int n_heads, float scale,
                          int max_tokens, void* d_dq, void* d_dk, void* d_dv, int64_t lddq, int64_t lddk, int64_t lddv,
                          void* stream);
int sst_sra_attn_bf16_profile_next(int backward, void* start, void* stop);
/* with a launch order of the windows (see sst_sra_attn_fwd_ord_f32) */
int sst_sra_attn_fwd_ord_bf16(const void* d_q, const void* d_k, const void* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, float scale, int max_tokens, void* d_o, int64_t ldo, float* d_lse,
                              void* stream);
int sst_sra_attn_bwd_ord_bf16(const void* d_q, const void* d_k, const void* d_v, const void* d_o, const void* d_do,
                              const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, float scale, int max_tokens, void* d_dq, void* d_dk, void* d_dv, int64_t lddq,
                              int64_t lddk, int64_t lddv, void* stream);
/* scaled cosine attention (cosine_msa.py:123-185) with bf16 storage: as sst_sra_attn_cos_{fwd,bwd}_f32 - normalisation in the
 * load prologue (the normalised rows rounded to bf16 before the products), d_head_scale [n_heads] fp32 in device memory =
 * 1 / clamp(tau, tau_min), gradients through the normalisation at the store, d_r [n_tokens, n_heads] fp32 = q^ . dq^ */
int sst_sra_attn_cos_fwd_bf16(const void* d_q, const void* d_k, const void* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, const float* d_head_scale, int max_tokens, void* d_o, int64_t ldo, float* d_lse,
                              void* stream);
int sst_sra_attn_cos_bwd_bf16(const void* d_q, const void* d_k, const void* d_v, const void* d_o, const void* d_do,
                              const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, const float* d_head_scale, int max_tokens, void* d_dq, void* d_dk, void* d_dv,
                              int64_t lddq, int64_t lddk, int64_t lddv, float* d_r, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels of the reduced-precision encoder layer: bf16 storage, fp32 parameters / statistics / arithmetic.
 *   sst_add_layernorm_fwd_bf16: y = LN(x + res) * w + b  (res may be NULL); d_sum (optional) = x + res for the backward
 *     pass; d_stats [m, 2] fp32 (mean, rstd); optional second output d_y_plus_pos = y + pos_table[pos_idx[row]] - the
 *     next layer's q / k input "x + positional embedding" (sst_basic_block_v2.py:58-60; pos_table: fp32 [P, c], the
 *     distinct rows of SSTInputLayerV2.get_pos_embed).
 *   sst_add_layernorm_bwd_bf16: d(x + res) from dy (+ dy2, optional second gradient arriving at y) and dweight / dbias
 *     (fp32).  Workspace: sst_add_layernorm_bwd_workspace_bytes(m, c).
 *   sst_cast_add_pos_bf16: out(bf16) = x (fp32 or bf16) [+ pos_table[pos_idx]]: entry of the bf16 stack.
 * ---------------------------------------------------------------------------------------------- */
int sst_add_layernorm_fwd_bf16(const void* d_x, const void* d_res, const float* d_weight, const float* d_bias, int64_t m,
                               int c, float eps, void* d_y, void* d_sum, float* d_stats, const float* d_pos_table,
                               const int32_t* d_pos_idx, void* d_y_plus_pos, void* stream);
int sst_add_layernorm_bwd_bf16(const void* d_dy, const void* d_dy2, const void* d_sum, const float* d_stats,
                               const float* d_weight, int64_t m, int c, void* d_dx, float* d_dweight, float* d_dbias,
                               void* d_workspace, void* stream);
int sst_cast_add_pos_bf16(const void* d_x, int x_is_bf16, int64_t m, int c, const float* d_pos_table,
                          const int32_t* d_pos_idx, void* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense products of the reduced-precision encoder layer (csrc/dense_bf16.hip): bf16 operands, fp32 accumulation.
 * Replace F.linear / nn.MultiheadAttention's in_proj / out_proj and linear1 - activation - linear2 of
 * mmdet3d/models/sst/sst_basic_block_v2.py:41-75, 104-126 (forward and autograd's data / weight gradients).
 *   sst_tall_linear_bf16: y[m, n] (bf16) = epilogue(x[m, k] w[n, k]^T + bias);  (k, n) in {(128,128), (128,256), (256,128)};
 *     w bf16 row-major [n][k] contiguous, bias fp32 or NULL.  epilogue: 0 = bias; 1 / 2 = GELU(erf) / ReLU, the bf16
 *     pre-activation also written to d_aux_out (may be NULL); 3 / 4 = multiply by GELU' / ReLU' of d_aux_in (the stored
 *     pre-activation): the data gradient through the activation; 5 = add the bf16 [m, n] tensor d_aux_in (a residual
 *     branch's gradient).  ldaux = row stride of d_aux_in / d_aux_out (elements).
 *   sst_wgrad_group_bf16: for every problem, out_w[p][128] (fp32) = a[m, p]^T b[m, 128] with p = 128 or 256
 *     (transpose_out: out_w[128][p] instead - the operands of a [128][256] weight swapped), and out_b = column sums of a
 *     (bias_side 1, p values) or of b (bias_side 2, 128 values).  All problems (<= 8) run in ONE launch + one reduction.
 * ---------------------------------------------------------------------------------------------- */
int sst_tall_linear_bf16(const void* d_x, int64_t ldx, const void* d_w, const float* d_bias, int64_t m, int k, int n,
                         int epilogue, const void* d_aux_in, void* d_aux_out, int64_t ldaux, void* d_y, int64_t ldy,
                         void* stream);
/* sst_tall_linear_ln_bf16: y = LayerNorm(x w^T + bias + res) - `norm(src + src2)` (sst_basic_block_v2.py:113-118) in the
 * epilogue of the projection that produces src2 (out_proj, linear2): n = 128, k = 128 or 256.  Also written: d_sum (bf16
 * [m, 128], row stride ldres like d_res; the backward pass's input; may be NULL), d_stats [m, 2] fp32 (mean, rstd), and, when
 * the positional arguments are given, d_y_plus_pos = y + pos_table[pos_idx[row]] (the next layer's q / k input). */
int sst_tall_linear_ln_bf16(const void* d_x, int64_t ldx, const void* d_w, const float* d_bias, int64_t m, int k,
                            const void* d_res, int64_t ldres, const float* d_ln_weight, const float* d_ln_bias, float eps,
                            void* d_y, void* d_sum, float* d_stats, const float* d_pos_table, const int32_t* d_pos_idx,
                            void* d_y_plus_pos, void* stream);
typedef struct sst_wgrad_problem_bf16 {
  const void* a;   /* bf16 [m, p], row stride lda */
  const void* b;   /* bf16 [m, 128], row stride ldb */
  int64_t lda, ldb, m;
  float* out_w;    /* device, fp32 [p][128] (or [128][p]) contiguous */
  float* out_b;    /* device, fp32, or NULL */
  int32_t p, bias_side, transpose_out, reserved;
} sst_wgrad_problem_bf16;
int64_t sst_wgrad_group_workspace_bytes(const sst_wgrad_problem_bf16* problems, int n);
int sst_wgrad_group_bf16(const sst_wgrad_problem_bf16* problems, int n, void* d_workspace, void* stream);
/* sst_cast_group_bf16: the bf16 copies of the fp32 master weights the reduced-precision mode computes with, any number of
 * them in one launch (what mmcv's Fp16OptimizerHook.copy_params_to_fp16 does after every optimizer step for
 * configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82).  Problem = rows x cols block of an fp32 matrix with row stride
 * ld_src -> bf16 [rows][cols] contiguous, or, with transpose, [cols][rows] (the operand of a data gradient).  problems is a
 * HOST array. */
typedef struct sst_cast_problem_bf16 {
  const float* src; /* device */
  void* dst;        /* device, bf16, rows * cols elements */
  int64_t ld_src;
  int32_t rows, cols, transpose, reserved;
} sst_cast_problem_bf16;
int sst_cast_group_bf16(const sst_cast_problem_bf16* problems, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a10/a14) row gather / scatter used by flat2window / window2flat / recover_bev.
 *   gather:  d_out[i, :] = idx[i] >= 0 ? d_src[idx[i], :] : fill
 *   scatter: d_out[idx[i], :] = d_src[i, :]      (idx unique; rows with idx < 0 skipped)
 * Replaces feat_3d[this_inds] = feat / feat[inds] (ops/sst/sst_ops.py:98, 124-125).
 * ---------------------------------------------------------------------------------------------- */
int sst_gather_rows_f32(const float* d_src, int64_t ld_src, const int32_t* d_idx, int64_t n_out, int c,
                        float fill, float* d_out, int64_t ld_out, void* stream);
/* out[i, :] = x[i, :] + table[idx[i], :]: q = k = feat + pos_embed of the first encoder layer (sst_basic_block_v2.py:62-66) with the
 * positional embedding kept as a table of the distinct in-window positions + a row index (sst_input_layer_v2.py:196-226 builds
 * the [M, C] tensor).  c % 4 == 0, 16-byte aligned rows. */
int sst_add_table_rows_f32(const float* d_x, int64_t ldx, const float* d_table, int64_t ld_table, const int32_t* d_idx, int64_t m,
                           int c, float* d_out, int64_t ld_out, void* stream);
int sst_scatter_rows_f32(const float* d_src, int64_t ld_src, const int32_t* d_idx, int64_t n_src, int c,
                         float* d_out, int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a13, §8 f1) row-wise pieces of an encoder layer around the attention core.
 *   sst_add_layernorm_fwd_f32: y = LayerNorm(x + res) * weight + bias per row (res may be NULL);
 *     replaces `src = self.norm1(src + src2)` (models/sst/sst_basic_block_v2.py:113-118).  d_sum (optional)
 *     receives x + res (saved for backward), d_stats [m,2] receives (mean, rstd).  1 <= c <= 512 (16-byte accesses
 *     when c % 4 == 0 and the operands are 16-byte aligned, 4-byte accesses otherwise: FSD's SIR layers have C = 133 ...).
 *   sst_add_layernorm_bwd_f32: d_dx = gradient w.r.t. (x + res) (same tensor for both addends);
 *     d_dweight / d_dbias [c] are overwritten with the column reductions.
 *   sst_colsum_f32: out[c] = sum over rows of x[m, c] (row stride ld) — bias gradients of the
 *     projections / FFN.  c % 4 == 0, c <= 1024.
 * ---------------------------------------------------------------------------------------------- */
/*   sst_add_layernorm_act_{fwd,bwd}_f32: the same with an activation behind the norm in the same pass, y = act(LayerNorm(
 *     x + res)) - "Linear -> LN -> GELU" of FSD's SIR layers and MLPs (voxel_encoder.py:628-650, sst_ops.py:334-361).
 *     act: 0 none, 1 GELU (erf form), 2 ReLU.  The backward takes dy at the activation's OUTPUT and recomputes the norm's
 *     output from the saved sum and statistics (hence d_bias). */
int sst_add_layernorm_act_fwd_f32(const float* d_x, const float* d_res, const float* d_weight, const float* d_bias,
                                  int64_t m, int c, float eps, int act, float* d_y, float* d_sum, float* d_stats,
                                  void* stream);
int sst_add_layernorm_act_bwd_f32(const float* d_dy, const float* d_sum, const float* d_stats, const float* d_weight,
                                  const float* d_bias, int act, int64_t m, int c, float* d_dx, float* d_dweight,
                                  float* d_dbias, void* d_workspace, void* stream);
int sst_add_layernorm_fwd_f32(const float* d_x, const float* d_res, const float* d_weight, const float* d_bias,
                              int64_t m, int c, float eps, float* d_y, float* d_sum, float* d_stats,
                              void* stream);
/* the same with a second output d_y_plus_pos[row] = y[row] + d_pos_table[d_pos_idx[row]] (table rows of c floats): the next
 * encoder layer's q / k input (sst_basic_block_v2.py:56-58: q = k = feat + pos) without an add pass; c % 4 == 0 */
int sst_add_layernorm_pos_fwd_f32(const float* d_x, const float* d_res, const float* d_weight, const float* d_bias, int64_t m,
                                  int c, float eps, float* d_y, float* d_sum, float* d_stats, const float* d_pos_table,
                                  const int32_t* d_pos_idx, float* d_y_plus_pos, void* stream);
int64_t sst_add_layernorm_bwd_workspace_bytes(int64_t m, int c);
int sst_add_layernorm_bwd_f32(const float* d_dy, const float* d_sum, const float* d_stats,
                              const float* d_weight, int64_t m, int c, float* d_dx, float* d_dweight,
                              float* d_dbias, void* d_workspace, void* stream);
int64_t sst_colsum_workspace_bytes(int64_t m, int c);
int sst_colsum_f32(const float* d_x, int64_t m, int c, int64_t ld, float* d_out, void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight / bias gradient of a tall linear layer y = x W^T + b (projections and FFN of an encoder layer,
 * models/sst/sst_basic_block_v2.py:104-126; VFE / SIR linears): dW[out,in] = dY[M,out]^T X[M,in] as a
 * split-K fp32 MFMA kernel, d_db[out] = column sums of dY (optional, NULL to skip).  1 <= out, in <= 4096.
 * Row strides ld_dy / ld_x in elements.  Workspace: sst_weight_grad_workspace_bytes(m, out, in).
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_weight_grad_workspace_bytes(int64_t m, int out, int in);
int sst_weight_grad_f32(const float* d_dy, const float* d_x, int64_t m, int out, int in, int64_t ld_dy,
                        int64_t ld_x, float* d_dw, float* d_db, void* d_workspace, void* stream);
/* The same for several problems (e.g. the five parameter gradients of an encoder layer): the split-K kernels one after
 * the other, ONE reduction launch for all.  dw [out, in] contiguous, db [out] or NULL. */
typedef struct sst_wgrad_problem_f32 {
  const float* dy;  /* [m, out], row stride ld_dy */
  const float* x;   /* [m, in], row stride ld_x */
  int64_t m, ld_dy, ld_x;
  float* dw;
  float* db;
  int32_t out, in;
  /* optional (both NULL or both set; exact-split group only, in == 128): the X operand is x + x_add_rows[x_add_index[token]] -
   * rows fp32 [P][in], index int32 [m]: the weight gradient of q | k = (feat + pos) W (sst_basic_block_v2.py:56-60) without
   * "feat + pos" in memory */
  const float* x_add_rows;
  const int32_t* x_add_index;
} sst_wgrad_problem_f32;
int64_t sst_weight_grad_group_workspace_bytes(const sst_wgrad_problem_f32* problems, int n);
int sst_weight_grad_group_f32(const sst_wgrad_problem_f32* problems, int n, void* d_workspace, void* stream);
/* The same gradients from the EXACT three-way bf16 split of both operands, six products on the bf16 matrix pipe with fp32
 * accumulation (csrc/wgrad_x6.hip; the arithmetic class of the fp32 kernel, see sst_tall_linear_epi_f32x6): ONE launch for all
 * problems (128 x 128 tiles of every dW x token slices) + one reduction.  All problems share m; out, in multiples of 128; row
 * strides % 4 == 0, 16-byte aligned operands; otherwise the workspace query returns SST_ERR_UNSUPPORTED (< 0) and the caller
 * uses sst_weight_grad_group_f32. */
int64_t sst_weight_grad_group_f32x6_workspace_bytes(const sst_wgrad_problem_f32* problems, int n);
int sst_weight_grad_group_f32x6(const sst_wgrad_problem_f32* problems, int n, void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (a11/a12, §8 f1) BatchNorm1d (+ ReLU) of the point-wise "Linear -> norm -> ReLU" layers of DynamicVFE / SIR
 * (models/voxel_encoders/utils.py:107-144; norm = BN1d or naiveSyncBN1d, ops/norm.py:28-86) over tall
 * x[n, c] (row strides in elements, c % 4 == 0, c <= 1024, 16-byte aligned rows).  Replaces the library's
 * batch_norm statistics / backward-reduce kernels and the separate ReLU passes.
 *   sst_bn_stats_f32: d_mean[c], d_var[c] = column mean and BIASED variance E[x^2] - mean^2 (fp64 sums).
 *   sst_bn_act_fwd_f32: y = act(x * scale[c] + shift[c]); act 0 = identity, 1 = ReLU
 *     (scale = weight * rsqrt(var + eps), shift = bias - mean * scale; the [c]-sized algebra, running
 *      statistics and the cross-rank averaging of naiveSyncBN stay on the host side).
 *   sst_bn_act_bwd_reduce_f32: d_sum_g[c] = sum g, d_sum_gxhat[c] = sum g * xhat with g = dy masked by the
 *     activation and xhat = (x - mean) * invstd  (= dbias, dweight of the affine norm).
 *   sst_bn_act_bwd_apply_f32: dx = scale * (g - coef_scale * coef_a[c] - xhat * coef_scale * coef_b[c])
 *     (training: coef_a = sum_g, coef_b = sum_gxhat, coef_scale = 1 / count; eval: coef_scale = 0).
 * Workspace: sst_bn_workspace_bytes(n, c) for the stats / reduce calls.
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_bn_workspace_bytes(int64_t n, int c);
int sst_bn_stats_f32(const float* d_x, int64_t n, int c, int64_t ld, float* d_mean, float* d_var,
                     void* d_workspace, void* stream);
int sst_bn_act_fwd_f32(const float* d_x, int64_t n, int c, int64_t ldx, const float* d_scale,
                       const float* d_shift, int act, float* d_y, int64_t ldy, void* stream);
int sst_bn_act_bwd_reduce_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                              const float* d_mean, const float* d_invstd, const float* d_scale,
                              const float* d_shift, int act, float* d_sum_g, float* d_sum_gxhat,
                              void* d_workspace, void* stream);
int sst_bn_act_bwd_apply_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                             const float* d_mean, const float* d_invstd, const float* d_scale,
                             const float* d_shift, const float* d_coef_a, const float* d_coef_b, float coef_scale,
                             int act, float* d_dx, int64_t lddx, void* stream);
/* Single-process training forward in one call: column moments of x, then d_out4 [4][c] = mean, invstd,
 * scale = weight * invstd, shift = bias - mean * scale, and running = (1 - factor) * running + factor * {mean,
 * var * n / (n - 1)} in place (nn.BatchNorm1d's bookkeeping; d_weight / d_bias / d_running_* may be NULL). */
int sst_bn_prepare_f32(const float* d_x, int64_t n, int c, int64_t ld, const float* d_weight, const float* d_bias,
                       float eps, float* d_running_mean, float* d_running_var, float factor, float* d_out4,
                       void* d_workspace, void* stream);
/*   sst_bn_prepare_tracked_f32: the same, and *d_num_batches_tracked += 1 (nn.BatchNorm1d's int64 counter; may be NULL)
 *     in the same launch. */
int sst_bn_prepare_tracked_f32(const float* d_x, int64_t n, int c, int64_t ld, const float* d_weight,
                               const float* d_bias, float eps, float* d_running_mean, float* d_running_var,
                               float factor, int64_t* d_num_batches_tracked, float* d_out4, void* d_workspace,
                               void* stream);
/* ------------------------------------------------------------------------------------------------
 * (f1) DynamicVFE's layer stack with its passes fused into their neighbours (voxel_encoder.py:185-298, utils.py:107-144:
 * per layer Linear(bias=False) -> BN -> ReLU -> scatter max -> cat with the pooled feature of the point's voxel).  Host side:
 * sst_amd/voxel_encoder.py FusedVFE2.
 *   sst_vfe_linear_moments_f32: first layer, y = x W^T for the K <= 16 decorated point channels (C % 4 == 0, C <= 256)
 *     and the block partials of the batch-norm moments of y in d_workspace (sst_bn_workspace_bytes(n, c)) from the same pass;
 *   sst_bn_prepare_from_partials_f32 / sst_bn_stats_from_partials_f32: the second halves of sst_bn_prepare_tracked_f32 /
 *     sst_bn_stats_f32 on those partials (the naiveSyncBN all-reduce of ops/norm.py:50-58 sits between the halves);
 *   sst_tall_linear_add_rows_f32x6 (below): second layer in its split-weight form;
 *   sst_segment_reduce_fwd_work_f32 with d_scale / d_shift: BN + ReLU of the last layer applied while it is pooled;
 *   sst_bn_act_pool_bwd_reduce_f32 / _apply_f32: sst_bn_act_bwd_reduce_f32 / _apply_f32 whose incoming gradient is
 *     d_dy (dense, may be NULL) + the gradient d_dpool [G, ldp] of the max pooling, routed to the recorded arg-max rows
 *     d_arg [G, c] through the point -> group map d_group [n] (negative: none): the dense matrix of the pooling's gradient
 *     (scatter_points_cuda.cu:135-179) is never written.
 * ---------------------------------------------------------------------------------------------- */
int sst_vfe_linear_moments_f32(const float* d_x, int64_t ldx, int64_t n, int k, const float* d_w, int64_t ldw, int c,
                               float* d_y, int64_t ldy, void* d_workspace, void* stream);
int sst_bn_prepare_from_partials_f32(int64_t n, int c, const float* d_weight, const float* d_bias, float eps,
                                     float* d_running_mean, float* d_running_var, float factor,
                                     int64_t* d_num_batches_tracked, float* d_out4, void* d_workspace, void* stream);
int sst_bn_stats_from_partials_f32(int64_t n, int c, float* d_mean, float* d_var, void* d_workspace, void* stream);
int sst_bn_act_pool_bwd_reduce_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                                   const float* d_mean, const float* d_invstd, const float* d_scale, const float* d_shift,
                                   int act, const int32_t* d_group, const int32_t* d_arg, const float* d_dpool, int64_t ldp,
                                   float* d_sum_g, float* d_sum_gxhat, void* d_workspace, void* stream);
int sst_bn_act_pool_bwd_apply_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                                  const float* d_mean, const float* d_invstd, const float* d_scale, const float* d_shift,
                                  const float* d_coef_a, const float* d_coef_b, float coef_scale, int act,
                                  const int32_t* d_group, const int32_t* d_arg, const float* d_dpool, int64_t ldp, float* d_dx,
                                  int64_t lddx, void* stream);
/* The residual tail of a sparse basic block (mmdet3d/ops/spconv/... sparse_block.py:127-139: bn2 -> += identity -> ReLU)
 * in the batch-norm passes: y = act(x * scale + shift + res).  The *_res_* entries are the calls above with the identity
 * branch d_res [n, c] (row stride ldr; NULL = none) inside the activation; the backward apply also writes the gradient of
 * the identity branch, d_dres = dy masked by the activation (NULL = not wanted). */
int sst_bn_act_res_fwd_f32(const float* d_x, int64_t n, int c, int64_t ldx, const float* d_res, int64_t ldr,
                           const float* d_scale, const float* d_shift, int act, float* d_y, int64_t ldy, void* stream);
int sst_bn_act_res_bwd_reduce_f32(const float* d_dy, const float* d_x, const float* d_res, int64_t n, int c,
                                  int64_t lddy, int64_t ldx, int64_t ldr, const float* d_mean, const float* d_invstd,
                                  const float* d_scale, const float* d_shift, int act, float* d_sum_g,
                                  float* d_sum_gxhat, void* d_workspace, void* stream);
int sst_bn_act_res_bwd_apply_f32(const float* d_dy, const float* d_x, const float* d_res, int64_t n, int c,
                                 int64_t lddy, int64_t ldx, int64_t ldr, const float* d_mean, const float* d_invstd,
                                 const float* d_scale, const float* d_shift, const float* d_coef_a,
                                 const float* d_coef_b, float coef_scale, int act, float* d_dres, int64_t lddres,
                                 float* d_dx, int64_t lddx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tall fp32 linear layer  y[m, n] (+)= x[m, k] W^T + bias  on the fp32 MFMA pipe with W resident in LDS
 * (projections / FFN of an encoder layer and their data gradients, models/sst/sst_basic_block_v2.py:41-75,
 * :104-126; replaces the library GEMM behind nn.Linear / F.linear for these shapes).
 *   trans_w = 0: d_w is [n][k] row-major (nn.Linear weight, forward).
 *   trans_w = 1: d_w is [k][n] row-major (data gradient dx = dy W of a layer whose weight is [out = k][in = n]).
 *   accumulate = 1: y += ... (residual / gradient accumulation rides on the product).
 * Supported: n == 128, k in {128, 256}; row strides in elements (multiples of 4, 16-byte aligned rows);
 * d_bias may be NULL.  Other shapes return SST_ERR_UNSUPPORTED (callers use the library GEMM for them).
 * ---------------------------------------------------------------------------------------------- */
int sst_tall_linear_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias,
                        int64_t m, int n, int k, int trans_w, int accumulate, float* d_y, int64_t ldy,
                        void* stream);
/* The same product (n == 128, k == 128) with the FFN's GELU (erf form, sst_basic_block_v2.py:116) in the epilogue:
 *   mode 0: y = x W^T + b and aux = gelu(y)   (linear1 + activation; both kept for the backward pass)
 *   mode 1: y = (x W [+ b]) * gelu'(aux)      (data gradient through linear2 times the activation's derivative at
 *                                              the pre-activation aux)
 * d_aux has the row stride ldy.  A 256-wide FFN is two calls on column halves of W / y / aux. */
/* sst_tall_linear_ln_f32: y = LayerNorm(x w^T + bias + res) in exact fp32 - `norm(src + src2)` (sst_basic_block_v2.py:113-118)
 * in the epilogue of the projection that produces src2 (out_proj, linear2; n = 128, k = 128 | 256; d_w [128][k] rows).  Also
 * written: d_sum ([m, 128], row stride ldres like d_res; may be NULL), d_stats [m, 2] (mean, rstd), and with the positional
 * arguments d_y_plus_pos = y + pos_table[pos_idx[row]].  sst_add_layernorm_bwd2_f32: the LayerNorm backward with an
 * optional SECOND upstream gradient d_dy2 (the one that arrives through y + pos), c = 128. */
int sst_tall_linear_ln_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias, int64_t m, int k,
                           const float* d_res, int64_t ldres, const float* d_ln_weight, const float* d_ln_bias, float eps,
                           float* d_y, float* d_sum, float* d_stats, const float* d_pos_table, const int32_t* d_pos_idx,
                           float* d_y_plus_pos, void* stream);
/* sst_tall_linear_epi_f32x3 / sst_tall_linear_ln_f32x3: the same two entry points (same arguments, fp32 tensors) with the
 * product evaluated on the bf16 matrix pipe from SPLIT operands, x w ~= x_hi w_hi + x_lo w_hi + x_hi w_lo, fp32 accumulation
 * (csrc/dense_f32x3.hip): ~1e-5 relative error per product - tighter than the TF32 the reference's nn.Linear ran on under
 * torch 1.8's default allow_tf32 = True on Ampere - at 3 / 16 of the fp32 pipe time.  Selected by SSTv2.set_precision('f32x3');
 * reported beside the exact-fp32 headline. */
int sst_tall_linear_epi_f32x3(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, int trans_w, const float* d_bias,
                              int64_t m, int k, int n, int epilogue, const float* d_aux_in, float* d_aux_out, int64_t ldaux,
                              float* d_y, int64_t ldy, void* stream);
int sst_tall_linear_ln_f32x3(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias, int64_t m, int k,
                             const float* d_res, int64_t ldres, const float* d_ln_weight, const float* d_ln_bias, float eps,
                             float* d_y, float* d_sum, float* d_stats, const float* d_pos_table, const int32_t* d_pos_idx,
                             float* d_y_plus_pos, void* stream);
/* sst_tall_linear_epi_f32x6 / sst_tall_linear_ln_f32x6 (csrc/dense_f32x6.hip): the same two entry points again, fp32 tensors,
 * the product evaluated on the bf16 matrix pipe from an EXACT three-way split of both operands (x = x0 + x1 + x2, 8 + 8 + 8
 * significand bits) with the six products x_i w_j, i + j <= 2, accumulated in fp32: what is dropped is below 2^-24 |x w| per
 * product, i.e. below the rounding of an fp32 FMA chain - the arithmetic class of the reference's nn.Linear in fp32
 * (sst_basic_block_v2.py:41-126), at 6 x 16 instead of 8 x 32 matrix-pipe cycles per 16 x 16 x 32 block.  (k, n) in
 * {(128,128), (128,256), (128,384), (256,128)}; the LayerNorm entry takes k = 128 only (SST_ERR_UNSUPPORTED for 256: the
 * caller runs epilogue 5 + sst_add_layernorm_act_fwd_f32).  Selected by SSTv2.set_precision('f32x6').
 * sst_tall_linear_epi2_f32x6: the same with a second input matrix d_x2 (same ldx) for the output columns >= x2_from_col (a
 * multiple of 128): q | k | v of an encoder layer in ONE launch, q = k = (x + pos) W, v = x W (sst_basic_block_v2.py:56-62). */
int sst_tall_linear_epi_f32x6(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, int trans_w, const float* d_bias,
                              int64_t m, int k, int n, int epilogue, const float* d_aux_in, float* d_aux_out, int64_t ldaux,
                              float* d_y, int64_t ldy, void* stream);
int sst_tall_linear_epi2_f32x6(const float* d_x, const float* d_x2, int x2_from_col, int64_t ldx, const float* d_w, int64_t ldw,
                               int trans_w, const float* d_bias, int64_t m, int k, int n, int epilogue, const float* d_aux_in,
                               float* d_aux_out, int64_t ldaux, float* d_y, int64_t ldy, void* stream);
/* y = x W^T + rows[row_index[r]] (negative index: row 0), (k, n) = (64, 128) or (64, 64): DynamicVFE's second layer on [point feature |
 * pooled feature of the point's voxel] (voxel_encoder.py:286-294: cat + Linear(128 -> 128)) as point_feats W[:, :64]^T +
 * (pooled W[:, 64:]^T)[voxel of the point].  sst_tall_linear_epi_f32x6 also takes (k, n) = (64, 128), (64, 64) and (128, 64). */
int sst_tall_linear_add_rows_f32x6(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, int64_t m, int k, int n,
                                   const float* d_rows, int64_t ldrows, const int32_t* d_row_index, float* d_y, int64_t ldy,
                                   void* stream);
/* The in-projection of an SRA encoder layer from x alone (sst_basic_block_v2.py:56-62: q = k = feat + pos, v = feat):
 * y[m, 384] = [(x + rows[index]) W[:256]^T | x W[256:]^T] + bias; W [384][128] (in_proj_weight), rows = positional table fp32
 * [P][128], index int32 [m].  One launch; "x + pos" is formed in registers. */
int sst_inproj_pos_f32x6(const float* d_x, int64_t ldx, const float* d_rows, const int32_t* d_index, const float* d_w, int64_t ldw,
                         const float* d_bias, int64_t m, float* d_y, int64_t ldy, void* stream);
int sst_tall_linear_ln_f32x6(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias, int64_t m, int k,
                             const float* d_res, int64_t ldres, const float* d_ln_weight, const float* d_ln_bias, float eps,
                             float* d_y, float* d_sum, float* d_stats, const float* d_pos_table, const int32_t* d_pos_idx,
                             float* d_y_plus_pos, void* stream);
int sst_add_layernorm_bwd2_f32(const float* d_dy, const float* d_dy2, const float* d_sum, const float* d_stats,
                               const float* d_weight, int64_t m, int c, float* d_dx, float* d_dweight, float* d_dbias,
                               void* d_workspace, void* stream);
/* ------------------------------------------------------------------------------------------------
 * One post-norm SRA encoder layer (sst_basic_block_v2.py:41-126, d_model 128, 8 heads, feed-forward 256) as ONE call forward
 * and ONE call backward (csrc/layer_exec.hip): the entry points above issued in the order of
 * sst_amd/sst_basic_block.py FusedEncoderLayerFn in its exact-split mode - nothing but the launch sequence moves from the
 * interpreter into the library (15 foreign calls per layer and step become 2).  All tensors fp32, row-major, contiguous with
 * the widths given; tok may be NULL (rows in window order), order may be NULL.
 *   forward : qkv [m, 384] = [(xp W_q^T | xp W_k^T) | x W_v^T] + b_in;  o, lse = SRA(q, k, v);
 *             y1 = LN1(x + o W_o^T + b_o) (s1 = the sum, st1 [m, 2] = mean, rstd; s1 may be NULL when nothing is kept);
 *             pre = y1 W_1^T + b_1 [m, 256], h = act(pre) (act 1 = GELU(erf), 2 = ReLU);  s2 = y1 + h W_2^T + b_2;
 *             y2 = LN2(s2), st2; y2p = y2 + pos_table[pos_idx] when pos_table != NULL (the next layer's x + pos).
 *   backward: from dy2 (and dy2p, may be NULL) the gradients of every parameter and dx (returned in ds1: d(x) with the
 *             gradient of xp folded in, xp = x + a constant); ds2, dpre, ds1, d_o, dqkv [m, 384] are scratch outputs of the
 *             widths 128, 256, 128, 128, 384; workspace: sst_encoder_layer_bwd_workspace_bytes(m, n_heads), 256-byte aligned.
 *   head_scale (last field of both): NULL = softmax(q k^T * scale); else scaled cosine attention (sst_sra_attn_cos_*_f32),
 *             [n_heads] floats = 1 / clamp(tau, tau_min) in device memory, and the backward writes cos_r [m, n_heads]
 *             (d head_scale[h] = colsum(cos_r)[h] / head_scale[h], taken by the caller).
 *   xp      : x + positional rows as a tensor, or NULL with (xpos_table, xpos_idx): the in-projection and the weight gradient of
 *             W_q | W_k then add the table rows to x on load (sst_inproj_pos_f32x6, x_add_rows of sst_wgrad_problem_f32) and
 *             no [m, 128] "x + pos" tensor exists; a chain of layers passes pos_table = NULL (no y2p) in that mode.
 *   wpack   : the weight images of the one-kernel tail (csrc/layer_tail_x6.hip: out-projection -> norm1 -> feed-forward -> norm2),
 *             sst_encoder_layer_wpack_bytes() bytes of device memory: formed by the forward call, read by the backward call.
 *   Launches: forward 4 (in-projection, attention core, weight images, tail), backward 5 (tail, attention core, the five
 *             weight gradients + their reduction - which also finishes the LayerNorm parameter gradients -, in-projection's data
 *             gradient).  dy1 (a scratch field of the round-5 sequence) is ignored.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sst_encoder_layer_fwd_args {
  int64_t m, n_windows;
  int32_t n_heads, act, max_tokens, impl;
  float eps, scale;
  const float *x, *xp;
  const float *w_in, *b_in, *w_out, *b_out, *w1, *b1, *w2, *b2, *n1w, *n1b, *n2w, *n2b;
  const int32_t *tok, *winoff, *order;
  const float* pos_table;
  const int32_t* pos_idx;
  float *qkv, *o, *lse, *y1, *s1, *st1, *pre, *h, *s2, *y2, *st2, *y2p;
  const float* head_scale;
  void* wpack;   /* sst_encoder_layer_wpack_bytes() bytes, written by the forward call, read by the backward call of the SAME layer call;
                    with w_out = w1 = w2 = NULL: already formed by sst_encoder_tail_pack_f32x6_many, only read */
  const float* xpos_table;     /* with xp == NULL: x + pos is formed on load from the positional table [P][128] ... */
  const int32_t* xpos_idx;     /* ... and the table row of every token (int32 [m]); then pos_table / y2p are normally NULL */
} sst_encoder_layer_fwd_args;
typedef struct sst_encoder_layer_bwd_args {
  int64_t m, n_windows;
  int32_t n_heads, act, max_tokens, impl;
  float eps, scale;
  const float *dy2, *dy2p;
  const float *x, *xp, *qkv, *o, *lse, *s1, *st1, *y1, *pre, *h, *s2, *st2;
  const float *w_in, *w_out, *w1, *w2, *n1w, *n2w;
  const int32_t *tok, *winoff, *order;
  float *ds2, *dpre, *ds1, *d_o, *dqkv;
  float *dw_in, *db_in, *dwo, *dbo, *dw1, *db1, *dw2, *db2, *dn1w, *dn1b, *dn2w, *dn2b;
  void* workspace;
  const float* head_scale;
  float* cos_r;
  float* dy1;    /* unused since round 6 (the gradient of y1 never leaves the registers of the tail kernel); may be NULL */
  const void* wpack;
  const float* xpos_table;     /* with xp == NULL (as in the forward call): the weight gradient of W_q | W_k reads x + table rows */
  const int32_t* xpos_idx;
} sst_encoder_layer_bwd_args;
int64_t sst_encoder_layer_bwd_workspace_bytes(int64_t m, int n_heads);
int64_t sst_encoder_layer_wpack_bytes(void);
int sst_encoder_layer_fwd_f32x6(const sst_encoder_layer_fwd_args* args, void* stream);
int sst_encoder_layer_bwd_f32x6(const sst_encoder_layer_bwd_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The part of that layer behind the attention core as ONE kernel per direction (csrc/layer_tail_x6.hip), exact-split
 * arithmetic, d_model 128, feed-forward 256 (sst_basic_block_v2.py:113-118 and their autograd): a wave carries 16 tokens
 * through out-projection -> + x -> norm1 -> linear1 -> act -> linear2 -> + y1 -> norm2 in registers; the weights pass
 * through LDS in 10 chunks.  All tensors fp32, contiguous, 16-byte aligned; widths: o, x, s1, y1, s2, y2, y2p, ds2, ds1,
 * d_o, dy2, dy2p 128; pre, h, dpre 256; st1, st2 [m, 2] (mean, rstd); w_out [128][128], w1 [256][128], w2 [128][256].
 *   pack    : `packed` (sst_encoder_tail_pack_bytes() bytes, device) = the three bf16 parts of the three weight matrices as the
 *             chunk images both kernels fetch with LDS-DMA; formed once per layer call (the weights of THIS forward pass: the
 *             backward pass must see the same buffer).
 *   forward : s1 = x + o W_o^T + b_o (s1 may be NULL), y1 = LN1(s1), pre = y1 W_1^T + b_1, h = act(pre) (act 1 = GELU(erf),
 *             2 = ReLU), s2 = y1 + h W_2^T + b_2 (s2 may be NULL), y2 = LN2(s2), y2p = y2 + pos_table[pos_idx] (all three NULL
 *             or all three given); biases may be NULL.
 *   backward: ds2 = LN2'(dy2 (+ dy2p, may be NULL)), dpre = (ds2 W_2) act'(pre), ds1 = LN1'(ds2 + dpre W_1), d_o = ds1 W_o;
 *             dn2w | dn2b | dn1w | dn1b [128] = the LayerNorm parameter gradients (m == 0: zeros).  workspace:
 *             sst_encoder_tail_bwd_workspace_bytes(m), 256-byte aligned.
 * The weight gradients of the three linears are the caller's (sst_weight_grad_group_f32x6 on (ds2, h), (dpre, y1), (ds1, o)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sst_encoder_tail_fwd_args {
  int64_t m;
  int32_t act, reserved;
  float eps, reserved_f;
  const float *o, *x;
  const void* packed;
  const float *b_out, *b1, *b2, *n1w, *n1b, *n2w, *n2b;
  const float* pos_table;
  const int32_t* pos_idx;
  float *s1, *st1, *y1, *pre, *h, *s2, *st2, *y2, *y2p;
} sst_encoder_tail_fwd_args;
typedef struct sst_encoder_tail_bwd_args {
  int64_t m;
  int32_t act, reserved;
  const float *dy2, *dy2p, *s2, *st2, *pre, *s1, *st1;
  const void* packed;
  const float *n1w, *n2w;
  float *ds2, *dpre, *ds1, *d_o;
  float *dn2w, *dn2b, *dn1w, *dn1b;
  void* workspace;
} sst_encoder_tail_bwd_args;
int64_t sst_encoder_tail_pack_bytes(void);
int sst_encoder_tail_pack_f32x6(const float* d_w_out, const float* d_w1, const float* d_w2, void* d_packed, void* stream);
/* the images of n layers in ONE launch (arrays of n device pointers in HOST memory; d_packed[i]: sst_encoder_tail_pack_bytes()
 * bytes each): what a stack of layers does once per forward pass (sst_v2.py:118-133 runs its blocks back to back);
 * sst_encoder_layer_fwd_f32x6 takes such images with w_out = w1 = w2 = NULL in its arguments */
int sst_encoder_tail_pack_f32x6_many(const float* const* d_w_out, const float* const* d_w1, const float* const* d_w2,
                                     void* const* d_packed, int n, void* stream);
int64_t sst_encoder_tail_bwd_workspace_bytes(int64_t m);
int sst_encoder_tail_fwd_f32x6(const sst_encoder_tail_fwd_args* args, void* stream);
int sst_encoder_tail_bwd_f32x6(const sst_encoder_tail_bwd_args* args, void* stream);

/* The same one-kernel tail in the reduced-precision mode (csrc/layer_tail_bf16.hip): activations bf16 (o, x, s1, y1, s2, y2, y2p,
 * ds2, ds1, d_o, dy2, dy2p [m, 128]; pre, h, dpre [m, 256]), statistics / biases / LayerNorm parameters / their gradients fp32;
 * `packed` = sst_encoder_tail_pack_bf16_bytes() bytes written by sst_encoder_tail_pack_bf16 from the fp32 master weights
 * (w_out [128][128], w1 [256][128], w2 [128][256]): the bf16 images of both directions.  Field meaning as sst_encoder_tail_*_args. */
typedef struct sst_encoder_tail_fwd_bf16_args {
  int64_t m;
  int32_t act, reserved;
  float eps, reserved_f;
  const void *o, *x;
  const void* packed;
  const float *b_out, *b1, *b2, *n1w, *n1b, *n2w, *n2b;
  const float* pos_table;
  const int32_t* pos_idx;
  void* s1;
  float* st1;
  void *y1, *pre, *h, *s2;
  float* st2;
  void *y2, *y2p;
} sst_encoder_tail_fwd_bf16_args;
typedef struct sst_encoder_tail_bwd_bf16_args {
  int64_t m;
  int32_t act, reserved;
  const void *dy2, *dy2p, *s2;
  const float* st2;
  const void *pre, *s1;
  const float* st1;
  const void* packed;
  const float *n1w, *n2w;
  void *ds2, *dpre, *ds1, *d_o;
  float *dn2w, *dn2b, *dn1w, *dn1b;
  void* workspace;
} sst_encoder_tail_bwd_bf16_args;
int64_t sst_encoder_tail_pack_bf16_bytes(void);
int sst_encoder_tail_pack_bf16(const float* d_w_out, const float* d_w1, const float* d_w2, void* d_packed, void* stream);
/* the images of n layers in one launch: host arrays of device pointers (a whole encoder stack per forward pass) */
int sst_encoder_tail_pack_bf16_many(const float* const* d_w_out, const float* const* d_w1, const float* const* d_w2,
                                    void* const* d_packed, int n, void* stream);
int64_t sst_encoder_tail_bwd_bf16_workspace_bytes(int64_t m);
int sst_encoder_tail_fwd_bf16(const sst_encoder_tail_fwd_bf16_args* args, void* stream);
int sst_encoder_tail_bwd_bf16(const sst_encoder_tail_bwd_bf16_args* args, void* stream);

/* The same layer in the reduced-precision mode (bf16 storage, fp32 accumulation / softmax / LayerNorm statistics, fp32 master
 * weights: what the reference's fp16 training of these layers - configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82 -
 * corresponds to), one call per direction: the launch sequence of sst_amd/bf16.py EncoderLayerBF16Fn issued from C (6 launches
 * forward, 10 + the grouped weight gradients backward; same kernels, same order, same bits).  d_model 128, feed-forward 256,
 * 8 heads.  Activations are bf16 [m, *]; wqk [256][128], wv / wout [128][128], w1 [256][128], w2 [128][256] are the bf16 copies
 * of the fp32 parameters (sst_cast_group_bf16), the *_t ones their transposes; biases, LayerNorm parameters, statistics
 * (st1, st2 [m, 2]) and lse [m, 8] are fp32.  Backward: dx / dxp (bf16) are the gradients of x and of xp = x + positional rows;
 * ds2, dpre, dy1, ds1, d_o, dqkv are scratch of widths 128, 256, 128, 128, 128, 384; parameter gradients fp32; workspace:
 * sst_encoder_layer_bwd_bf16_workspace_bytes(m), 256-byte aligned.  head_scale (last fields): NULL = standard attention, else
 * scaled cosine attention (sst_sra_attn_cos_*_bf16) and the backward writes cos_r [m, 8] fp32. */
typedef struct sst_encoder_layer_fwd_bf16_args {
  int64_t m, n_windows;
  int32_t n_heads, act, max_tokens, reserved;
  float eps, scale;
  const void *x, *xp;
  const void *wqk, *wv, *wout, *w1, *w2;
  const float *b_in, *b_out, *b1, *b2, *n1w, *n1b, *n2w, *n2b;
  const int32_t *tok, *winoff, *order;
  const float* pos_table;
  const int32_t* pos_idx;
  void *qk, *v, *o;
  float* lse;
  void *y1, *s1;
  float* st1;
  void *pre, *h, *s2;
  float* st2;
  void *y2, *y2p;
  const float* head_scale;
  const void* wpack;   /* sst_encoder_tail_pack_bf16 images of this layer's out_proj / linear1 / linear2 (fp32 masters) */
} sst_encoder_layer_fwd_bf16_args;
typedef struct sst_encoder_layer_bwd_bf16_args {
  int64_t m, n_windows;
  int32_t n_heads, act, max_tokens, reserved;
  float eps, scale;
  const void *dy2, *dy2p;
  const void *x, *xp, *qk, *v, *o;
  const float* lse;
  const void* s1;
  const float* st1;
  const void *y1, *pre, *h, *s2;
  const float* st2;
  const void *wqk_t, *wv_t, *wout_t, *w1_t, *w2_t;
  const float *n1w, *n2w;
  const int32_t *tok, *winoff, *order;
  void *ds2, *dpre, *dy1, *ds1, *d_o, *dqkv, *dxp, *dx;
  float *dw_in, *db_in, *dwo, *dbo, *dw1, *db1, *dw2, *db2, *dn1w, *dn1b, *dn2w, *dn2b;
  void* workspace;
  const float* head_scale;
  float* cos_r;
  const void* wpack;   /* as in the forward arguments */
} sst_encoder_layer_bwd_bf16_args;
int64_t sst_encoder_layer_bwd_bf16_workspace_bytes(int64_t m);
int sst_encoder_layer_fwd_bf16(const sst_encoder_layer_fwd_bf16_args* args, void* stream);
int sst_encoder_layer_bwd_bf16(const sst_encoder_layer_bwd_bf16_args* args, void* stream);

/* sst_tall_linear_epi_f32 (csrc/dense_f32.hip): y[m, n] = epilogue(x[m, k] W^T + bias), exact fp32 (v_mfma_f32_16x16x4_f32),
 * the whole weight matrix resident in LDS; (k, n) in {(128,128), (128,256), (256,128)}.  trans_w = 0: d_w holds W as [n][k]
 * rows (F.linear's weight); trans_w = 1: d_w holds [k][n] rows - the data gradient dy[m, out] w[out, in] of a layer with
 * parameter w (sst_basic_block_v2.py:41-75, 104-126 and their autograd).  epilogue: 0 = bias; 1 / 2 = GELU(erf) / ReLU with
 * the pre-activation also written to d_aux_out (may be NULL); 3 / 4 = multiply by GELU' / ReLU' of d_aux_in; 5 = add
 * d_aux_in (may alias d_y: in-place accumulation of a residual branch's gradient). */
int sst_tall_linear_epi_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, int trans_w, const float* d_bias,
                            int64_t m, int k, int n, int epilogue, const float* d_aux_in, float* d_aux_out, int64_t ldaux,
                            float* d_y, int64_t ldy, void* stream);
int sst_tall_linear_gelu_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias,
                             int64_t m, int n, int k, int trans_w, int mode, float* d_aux, float* d_y, int64_t ldy,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py `roofline`): HIP events attached to ONE launch of the register-resident SRA forward
 * kernel.  sst_sra_attn_profile_next_fwd(start, stop) arms the next sst_sra_attn_fwd_f32 call (impl 0, windows
 * <= 144 tokens): the kernel is launched with hipExtLaunchKernelGGL so that `start` / `stop` carry the kernel's own
 * begin / end timestamps on the launch stream.  Events come from sst_event_create (plain hipEventCreate).
 * ---------------------------------------------------------------------------------------------- */
void* sst_event_create(void);
void sst_event_destroy(void* ev);
float sst_event_elapsed_ms(void* start, void* stop); /* synchronises on `stop`; < 0 on error */
int sst_sra_attn_profile_next_fwd(void* start, void* stop);
/* the same for the one-pass backward kernel (the next sst_sra_attn_bwd_f32 call with impl 0); both hooks are
 * one-shot and local to the calling thread */
int sst_sra_attn_profile_next_bwd(void* start, void* stop);

/* ------------------------------------------------------------------------------------------------
 * (§8 f2) Connected components of the graph "same sample and xy distance < dist" over n points — FSD's
 * ClusterAssigner.  Replaces find_connected_componets (models/detectors/single_stage_fsd.py:45-68): dense N x N
 * distance matrix, `.cpu()`, scipy.sparse.csgraph.connected_components per sample, running label base.
 *   d_points [n, >= 2] fp32 (x, y first; row stride ld elements), d_batch [n] int32 sample index (samples stored
 *   one after the other, as the sorted-unique that produces the centres leaves them).
 *   d_labels [n] int32: components numbered by their smallest point index (scipy's order of first appearance;
 *   with contiguous samples = the reference's per-sample numbering + running base).  Bit-exact: the edge test
 *   uses the reference's fp32 operations (sub, mul, add, sqrt, <).  d_num_components (optional, device int32).
 * Workspace: sst_connected_components_workspace_bytes(n).
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_connected_components_workspace_bytes(int64_t n);
int sst_connected_components_xy_f32(const float* d_points, int64_t ld, const int32_t* d_batch, int64_t n, float dist,
                                    int32_t* d_labels, int32_t* d_num_components, void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (§8 f3) Dynamic point pool — the RoI extractor in front of FSD's second-stage SIR layers.  Replaces
 * dynamic_point_pool_ext.forward(rois, pts, extra_wlh, max_inbox_point, out_pts_idx, out_roi_idx, out_pts_feats)
 * (mmdet3d/ops/dynamic_point_pool_op.py:36) and dynamic_point_pool_ext.dynamic_point_pool_mixed_gpu (:86-88); that
 * extension (TorchEx) is not in the reference tree: PARITY UNPINNED beyond the box convention of
 * ops/roiaware_pool3d/src/points_in_boxes_cuda.cu:24-50 and the invariants asserted in
 * models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:96-105.
 *   d_rois [n_rois, 7] fp32 (x, y, z of the bottom centre, w, l, h, rz); d_pts [n_pts, >= 3] fp32, row stride ld_pts.
 *   d_rois_batch / d_pts_batch: both NULL (one sample) or int32 sample indices (a pair needs equal indices).
 *   extra_wlh: HOST pointer to 3 floats added to (w, l, h) for the membership test.
 *   Outputs (caller-allocated, max_all_pts rows; rows >= *d_num_out are left untouched): d_out_pts_idx,
 *   d_out_roi_idx int64; d_out_feats [max_all_pts, 13] = point xyz, box-frame xyz, distances to the six faces of the
 *   un-enlarged box (+x, +y, +z side sums give l, w, h), 1.0 if the point lies only in the enlarged margin.
 *   Deterministic: pairs sorted by (RoI, point index); a RoI keeps its first max_inbox_point points, the output
 *   its first max_all_pts pairs (the reference's atomics make the surviving subset and the order race-dependent).
 *   d_num_out: device int64.  Workspace: sst_dynamic_point_pool_workspace_bytes(n_rois, n_pts).
 * ---------------------------------------------------------------------------------------------- */
int64_t sst_dynamic_point_pool_workspace_bytes(int64_t n_rois, int64_t n_pts);
int sst_dynamic_point_pool_f32(const float* d_rois, const int32_t* d_rois_batch, int64_t n_rois, const float* d_pts,
                               int64_t ld_pts, const int32_t* d_pts_batch, int64_t n_pts, const float* extra_wlh,
                               int max_inbox_point, int64_t max_all_pts, int64_t* d_out_pts_idx,
                               int64_t* d_out_roi_idx, float* d_out_feats, int64_t* d_num_out, void* d_workspace,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * (§8 f4, first part) Sparse 3-D convolution: rulebook + convolution.  Replaces the reference's vendored spconv for
 * 3-D int32 indices: getIndicePair<3> (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:26-150; kernel-offset and
 * position arithmetic include/spconv/geometry.h:24-151) and indiceConv / indiceConvBackward (spconv_ops.h:256-446).
 * All index tensors are int32; d_coors rows are (batch, z, y, x), contiguous, 16-byte aligned; shapes / ksize /
 * stride / padding / dilation are HOST arrays of 3 int32 (z, y, x).  Kernel offset k = (kz * ksize_y + ky) * ksize_x
 * + kx (the `offset` of geometry.h:66-70); K = ksize_z * ksize_y * ksize_x <= 4096.
 *
 * The rulebook is two dense maps, in2out [K, n] and out2in [K, m] (-1 = no partner), plus the reference's pair lists
 * (indice_pairs [K, 2, n] filled with -1 behind the valid entries, indice_num [K]; pairs ordered by input row).
 *   sst_spconv_candidates_i32: regular / transposed convolution, rows [K * n + 1, 4] = output voxel (b, z, y, x) touched
 *     by (offset k, input row j) at row k * n + j, or four -1; the last row is always invalid.  Feed them to
 *     sst_unique_rows(invalid_if_negative = 1): the sorted groups 1.. are the output voxels in ascending (b, z, y, x)
 *     order (the order of the reference's GPU path, torch::_unique of the linear indices, spconv_ops.h:128), and
 *     sst_spconv_inverse_to_map_i32 turns its inverse into in2out (group - 1).
 *   sst_spconv_subm_map_i32: submanifold convolution (stride 1, padding ksize / 2, spconv_ops.h:74-77): outputs =
 *     inputs.  d_sorted_keys / d_perm: ukeys and perm of sst_unique_rows over the n input rows with mins = 0,
 *     extents = (batch, shape) and invalid_if_negative = 0 (key = 1 + linear index).
 *   sst_spconv_invert_map_i32: out2in from in2out.   sst_spconv_pair_lists_i32: pair lists from in2out.
 *   sst_spconv_gather_gemm_f32: Y[r, :] = sum_k X[map[k][r], :] W[k] (+ bias), r < m; W[k] is [cin, cout] row-major, or
 *     [cout, cin] with trans_w (data gradient on the forward weights).  Forward: map = out2in; data gradient and
 *     inverse convolution: map = in2out.  Every row of Y is written.  form: 0 = automatic, 1 = every row of a 64 x 128
 *     tile times every offset present in the tile, 2 = MFMAs over the compacted rows that have a partner (K <= 32,
 *     W not transposed; best when few of the K offsets are populated).
 *   sst_spconv_wgrad_f32: dW[k] = sum over the pairs p < num[k] of X[pairs[k][x_side][p]]^T dY[pairs[k][1 - x_side][p]];
 *     pair_ld = row length of the pair lists (n); total_pairs = sum of d_num if the caller knows it on the host (it bounds
 *     the launch and the workspace), -1 otherwise.  Workspace: sst_spconv_wgrad_workspace_bytes (same arguments).
 * ---------------------------------------------------------------------------------------------- */
int sst_spconv_candidates_i32(const int32_t* d_coors, int64_t n, const int32_t* in_shape, const int32_t* out_shape,
                              const int32_t* ksize, const int32_t* stride, const int32_t* padding,
                              const int32_t* dilation, int transpose, int32_t* d_rows, void* stream);
int sst_spconv_inverse_to_map_i32(const int32_t* d_inverse, int64_t total, int32_t* d_in2out, void* stream);
/*   Dense-grid builders of the SAME rulebooks (getIndicePair<3>, spconv_ops.h:26-150; identical numbering): a cell -> row
 *     grid over batch x shape replaces the sort / binary search when the grid fits (cells <= 2^31).
 *     sst_spconv_grid_subm_i32: d_grid [batch * prod(shape)] int32 scratch -> in2out AND out2in [K, n].
 *     sst_spconv_grid_conv_count_i32: regular / transposed convolution, pass 1: flags the output cells, scans them,
 *     *d_num_out = number of output voxels (device int32: the caller reads it back to size pass 2's outputs);
 *     workspace sst_spconv_grid_conv_workspace_bytes(batch * prod(out_shape)), handed on to
 *     sst_spconv_grid_conv_maps_i32: d_outids [m, 4] (b, z, y, x) ascending, in2out [K, n], out2in [K, m]. */
int sst_spconv_grid_subm_i32(const int32_t* d_coors, int64_t n, int batch, const int32_t* shape, const int32_t* ksize,
                             const int32_t* dilation, int32_t* d_grid, int32_t* d_in2out, int32_t* d_out2in,
                             void* stream);
int64_t sst_spconv_grid_conv_workspace_bytes(int64_t cells);
int sst_spconv_grid_conv_count_i32(const int32_t* d_coors, int64_t n, int batch, const int32_t* in_shape,
                                   const int32_t* out_shape, const int32_t* ksize, const int32_t* stride,
                                   const int32_t* padding, const int32_t* dilation, int transpose, void* d_workspace,
                                   int32_t* d_num_out, void* stream);
int sst_spconv_grid_conv_maps_i32(const int32_t* d_coors, int64_t n, int batch, const int32_t* in_shape,
                                  const int32_t* out_shape, const int32_t* ksize, const int32_t* stride,
                                  const int32_t* padding, const int32_t* dilation, int transpose, void* d_workspace,
                                  int64_t m, int32_t* d_outids, int32_t* d_in2out, int32_t* d_out2in, void* stream);
int sst_spconv_subm_map_i32(const int32_t* d_coors, int64_t n, const int32_t* shape, const int32_t* ksize,
                            const int32_t* dilation, const uint64_t* d_sorted_keys, const uint32_t* d_perm,
                            int32_t* d_in2out, void* stream);
int sst_spconv_invert_map_i32(const int32_t* d_in2out, int kvol, int64_t n, int64_t m, int32_t* d_out2in,
                              void* stream);
int64_t sst_spconv_pair_lists_workspace_bytes(int kvol, int64_t n);
int sst_spconv_pair_lists_i32(const int32_t* d_in2out, int kvol, int64_t n, int32_t* d_pairs, int32_t* d_num,
                              void* d_workspace, void* stream);
int sst_spconv_gather_gemm_f32(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol,
                               const float* d_w, int cin, int cout, int trans_w, const float* d_bias, float* d_y,
                               int64_t ldy, int form, void* stream);
/*   sst_spconv_maxpool_{fwd,bwd}_f32: indiceMaxPool / indiceMaxPoolBackward (include/spconv/pool_ops.h:24-97,
 *     src/maxpool.cc:22-63) on the maps: y[i] = max(0, max_k x[out2in[k][i]]) (the reference's zero-filled start);
 *     dx[j] = sum of dy[i] over the outputs i = in2out[k][j] whose value equals x[j] (ties all receive). */
int sst_spconv_maxpool_fwd_f32(const float* d_x, int64_t ldx, const int32_t* d_out2in, int64_t m, int kvol, int c,
                               float* d_y, int64_t ldy, void* stream);
int sst_spconv_maxpool_bwd_f32(const float* d_x, int64_t ldx, const float* d_y, int64_t ldy, const float* d_dy,
                               int64_t lddy, const int32_t* d_in2out, int64_t n, int kvol, int c, float* d_dx,
                               int64_t lddx, void* stream);
/*   sst_spconv_conv_os_f32: the same contraction as sst_spconv_gather_gemm_f32 (indiceConv and the data gradient of
 *     indiceConvBackward, spconv_ops.h:256-446) as an output-stationary implicit GEMM: W[k] packed into MFMA-fragment
 *     order (d_workspace, sst_spconv_conv_os_workspace_bytes) and staged through LDS, partner rows gathered straight into
 *     MFMA operands, XCD-aware tile numbering (csrc/spconv_os.hip).  K <= 32, cin % 4 == 0, ldx % 4 == 0, d_x 16-byte
 *     aligned; SST_ERR_UNSUPPORTED otherwise (use sst_spconv_gather_gemm_f32).  trans_w as above.  tile_cfg: 0 = tile
 *     shape chosen from m and cout, else 10 * (column tiles of 16: 4 | 8) + (16-row blocks per wave: 1 | 2).
 *     d_tile_order (may be NULL): a permutation of the ceil(m / tile_rows) row tiles, the order in which they are launched -
 *     heaviest first evens out the end of the launch; tile_rows = sst_spconv_conv_os_tile_rows(m, cout, tile_cfg) (64 or
 *     128), and sst_spconv_os_tile_work_i32 writes work[tile] = populated (16-row block, offset) slots of the tile, the key
 *     to sort by (descending).  The result does not depend on the order. */
int64_t sst_spconv_conv_os_workspace_bytes(int kvol, int cin, int cout);
int sst_spconv_conv_os_tile_rows(int64_t m, int cout, int tile_cfg);
int sst_spconv_os_tile_work_i32(const int32_t* d_map, int64_t m, int kvol, int tile_rows, int32_t* d_work, void* stream);
int sst_spconv_conv_os_f32(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                           int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy,
                           int tile_cfg, const int32_t* d_tile_order, void* d_workspace, void* stream);
/*   sst_spconv_conv_os_f32x3: sst_spconv_conv_os_f32 in SPLIT precision - every fp32 product as three bf16 products of split
 *     operands (x w ~= x_hi w_hi + x_lo w_hi + x_hi w_lo) on v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 in memory
 *     (csrc/spconv_os_x3.hip; ~1e-5 of the output scale).  Same arguments, workspace size and launch order; tile_cfg = 0. */
int sst_spconv_conv_os_f32x3(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                             int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy, int tile_cfg,
                             const int32_t* d_tile_order, void* d_workspace, void* stream);
/*   sst_spconv_conv_os_f32x6: the same contraction from the EXACT three-way bf16 split of both operands, six products per
 *     fp32 product on v_mfma_f32_16x16x32_bf16 with fp32 accumulation in two accumulator sets (csrc/spconv_os_x6.hip; the
 *     arithmetic class of the fp32 kernel: error vs float64 <= 2 x its error, tests/test_gpu_spconv.py).  Same arguments and
 *     launch order; workspace: sst_spconv_conv_os_f32x6_workspace_bytes (6 bytes per packed weight element); tile_cfg = 0. */
int64_t sst_spconv_conv_os_f32x6_workspace_bytes(int kvol, int cin, int cout);
int sst_spconv_conv_os_f32x6(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                             int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy, int tile_cfg,
                             const int32_t* d_tile_order, void* d_workspace, void* stream);
/* sst_spconv_conv_os_rows_f32x6: the same call, told the size of its workspace.  With
 * sst_spconv_conv_os_f32x6_workspace_bytes_rows(kvol, cin, cout, m) bytes (packed weights + partial tiles) a THIN level - a few
 * thousand rows: 30-200 (row tile, column group) units for 256 CUs, each walking all 27 offsets alone - has its offsets dealt
 * out over up to 8 workgroups per unit, whose partial tiles are added in a fixed order by a second launch (deterministic; same
 * results up to the association of the per-offset sums).  d_workspace 256-byte aligned. */
int64_t sst_spconv_conv_os_f32x6_workspace_bytes_rows(int kvol, int cin, int cout, int64_t m);
int sst_spconv_conv_os_rows_f32x6(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                                  int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy, int tile_cfg,
                                  const int32_t* d_tile_order, void* d_workspace, int64_t workspace_bytes, void* stream);
/*   sst_spconv_wgrad_os_f32: the same filter gradient as sst_spconv_wgrad_f32 (indiceConvBackward, spconv_ops.h:359-446)
 *     with the gathered rows staged through LDS transposed, 64 x 64 blocks of dW[k], 2048-pair chunks (csrc/spconv_os.hip).
 *     cin % 4 == 0, cout % 4 == 0, row strides % 4 == 0, 16-byte aligned operands; SST_ERR_UNSUPPORTED otherwise. */
int64_t sst_spconv_wgrad_os_workspace_bytes(int kvol, int64_t pair_ld, int64_t total_pairs, int cin, int cout);
int sst_spconv_wgrad_os_f32(const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                            int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                            int cout, float* d_dw, void* d_workspace, void* stream);
/*   sst_spconv_wgrad_os_f32x6: the same gradient from the exact three-way bf16 split of both gathered operands (six products on the
 *   bf16 matrix pipe, fp32 accumulation, two accumulator sets; csrc/spconv_os.hip sp_wgrad_os_x6_k): the filter-gradient half of
 *   the 'f32x6' convolution precision.  Same arguments and workspace as sst_spconv_wgrad_os_f32. */
int sst_spconv_wgrad_os_f32x6(const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                            int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                            int cout, float* d_dw, void* d_workspace, void* stream);
int64_t sst_spconv_wgrad_workspace_bytes(int kvol, int64_t pair_ld, int64_t total_pairs, int cin, int cout);
int sst_spconv_wgrad_f32(const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                         int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                         int cout, float* d_dw, void* d_workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SST_AMD_H */
