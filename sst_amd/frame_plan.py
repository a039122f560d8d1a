"""Index plan of a frame batch without host round trips (csrc/frame_plan.hip).

The piecewise path (``Voxelization`` -> ``DynamicVFE.scatter_plan`` -> ``SSTInputLayerV2.build_plan``) mirrors the
reference's module boundaries, and every boundary hands over a tensor whose first dimension is a device-side count
(voxels after the sorted-unique, survivors and windows after the drop): three read-backs and ~150 small launches per
step.  ``FramePlan`` builds the same plan - the point -> voxel grouping DynamicVFE reduces over (with the reference's
"first voxel of every sample" quirk), the kept voxels in window-major order, the window CSR of both partitions, the
positional-embedding rows - from upper-bound sized buffers, keeps the counts on the device, and reads them ONCE in
``finalize()``, which the caller invokes after it has queued the voxel encoder's feature kernels.

Semantics: kept-voxel set, window membership and window order are those of ``SSTInputLayerV2`` with
``window_major=True``; with ``shuffle_voxels=False`` the drop is by ascending voxel index (tests compare the two
paths bit for bit), with ``shuffle_voxels=True`` a uniformly random subset of every over-full window survives, as
with the reference's ``randperm`` (sst_input_layer_v2.py:93-97, 128-148).
"""
import torch

from . import _lib
from . import kernels as K


class FramePlan(object):

    def __init__(self):
        # (the plan does NOT keep the voxel_info it hands out: info['voxel_feats'] carries the autograd graph of the voxel encoder,
        # whose nodes hold this plan - plan -> info -> graph -> plan was a reference cycle, and every step's plan, index buffers
        # and gathered features, ~65 MB at the bench size, waited for Python's cyclic collector instead of dying with the step)
        self.want_pos_rows = True   # materialise the [M, C] positional tensors (the fp32 encoder layers add them to x)

    # -- the interface DynamicVFE.forward uses of a scatter plan ------------------------------------------------
    def reduce(self, feats, mode):
        """Segmented reduce of point features over the kept voxels: [n_upper, C], rows >= M (device count) are
        never written or read."""
        return K.segment_reduce(feats, self.groups, mode, group_index=self.gidx, inverse=self.coors_map,
                                m_limit=self.d_counts[0:1])

    def raw_max(self, feats, scale_shift=None):
        """(max, arg-max rows) over the kept voxels without an autograd node; scale_shift: values are read as
        relu(x * scale + shift) (vfe_fused.FusedVFE2)"""
        return K._segment_reduce_fwd(feats, self.groups.perm, self.groups.offsets, self.gidx.numel(), K.REDUCE['max'], True,
                                     self.gidx, self.d_counts[0:1], self.groups, scale_shift)

    def group_sum(self, part):
        """Gradient of the hand-back ``pooled[coors_map]`` (points of a discarded group read row 0): the CSR sum over the kept
        voxels, plus the rows of the discarded groups added to row 0 - in a fixed order, where an index_add with float
        atomics is neither reproducible nor fast on the crowded voxels of a real sweep."""
        dg = self.reduce(part, 'sum')
        m = getattr(self, 'num_voxels', None)
        if m is None:
            m = int(self.d_counts[0].item())
        if m < dg.size(0):
            dg[m:].zero_()
        return K.add_group_rows_to_row0(dg, part, self.groups, self.dropped_gidx)

    @property
    def voxel_coors(self):
        return self.vcoors

    def _release_count_slot(self):
        """hand the pinned size words back to the planner's free list (their values have been read, or nobody will)"""
        home, slot = self.__dict__.pop('_slot_home', None), (self.__dict__.pop('h_counts', None), self.__dict__.pop('counts_ready', None))
        if home is not None and slot[0] is not None and len(home) < 64:
            home.append(slot)

    def __del__(self):
        try:
            if '_slot_home' in self.__dict__:
                # unread plan (built ahead and dropped): its copy may still be in flight - wait for it before the words are reused
                self.counts_ready.synchronize()
                self._release_count_slot()
        except Exception:
            pass

    # -----------------------------------------------------------------------------------------------------------
    def finalize(self, voxel_feats, input_layer, feat_dim=None):
        """Read the sizes (the only host synchronisation of the plan) and return the ``voxel_info`` dictionary the
        SST backbone consumes: surviving voxel features in window-major order, their coordinates, the window CSR and
        the positional embedding of both partitions."""
        # the sizes were copied to pinned host memory right behind the plan's kernels (build()): waiting for THAT copy does
        # not wait for anything queued after it (the voxel encoder of this step, or - when the plan was built ahead, behind
        # the previous step's backward pass - nothing at all)
        counts = self.__dict__.get('_counts')
        if counts is None:               # a second finalize() of the same plan (activation checkpointing around voxel_info,
            self.counts_ready.synchronize()   # extract_feat after extract_voxel_feats) reuses what the first one read
            counts = self._counts = self.h_counts.tolist()
            self._release_count_slot()
        m, m_keep, n_win, t_max = counts[0], counts[1], (counts[2], counts[3]), (counts[4], counts[5])
        self.num_voxels, self.num_kept = m, m_keep
        dev = voxel_feats.device
        feat_index = self.feat_index[:m_keep]
        from .sst_input_layer import VoxelInfo
        info = VoxelInfo({'voxel_feats': GatherRows.apply(voxel_feats, feat_index, self.feat_index_i32[:m_keep]),
                          'voxel_coors': self.out_coors[:m_keep], 'voxel_keep_inds': feat_index})
        cap = self.max_tokens_cap
        tok0 = torch.arange(m_keep, dtype=torch.int32, device=dev)
        toks = (tok0, self.tok1)
        winoffs = (self.winoff0, self.winoff1)
        dim = voxel_feats.size(1) if feat_dim is None else feat_dim
        table = input_layer.pos_table_cached(dim, voxel_feats.dtype, dev)
        for i in range(2):
            info[f'sra_plan_shift{i}'] = K.WindowPlan(toks[i], winoffs[i], n_win[i], m_keep,
                                                      min(cap, max(1, t_max[i])), rows_in_window_order=(i == 0))
            info[f'pos_index_shift{i}'] = (self.posidx0, self.posidx1)[i][:m_keep]
        info['pos_table'] = table
        info['batch_size'] = self.batch_size
        # the [M, C] positional tensors (only the per-layer path adds them as tensors) and the reference-style entries
        # (flat2win dictionaries, padded positional tensors, key masks ...) are formed when somebody reads them
        info.defer([f'pos_embed_shift{i}' for i in range(2)], lambda d: d.update(
            {f'pos_embed_shift{i}': K.gather_rows(d['pos_table'], d[f'pos_index_shift{i}']) for i in range(2)}))
        if input_layer.reference_outputs or input_layer.debug:
            input_layer.defer_reference_entries_of_kept(info)
            if input_layer.shuffle_voxels:
                # voxel_feats_out = voxel_feats_in[shuffle_inds][voxel_keep_inds] (sst_input_layer_v2.py:93-97, 150-226): the
                # plan's row index already maps output rows to input rows, so the shuffle part is the identity
                # (the closure must not hold the plan: plan -> info -> closure -> plan would be a reference cycle, and every
                # step's plan - ~65 MB of index buffers - would wait for the cyclic collector instead of dying with the step)
                info.defer(['shuffle_inds'], lambda d, _m=m, _dev=dev: d.update(
                    shuffle_inds=torch.arange(_m, dtype=torch.int64, device=_dev)))
        return info


class GatherRows(torch.autograd.Function):
    """rows ``index`` of a [N, C] fp32 tensor (index entries unique); backward = zero-fill + row scatter (no atomics)."""

    @staticmethod
    def forward(ctx, src, index_i64, index_i32):
        ctx.save_for_backward(index_i32)
        ctx.rows = src.size(0)
        return K.gather_rows(src.contiguous(), index_i32)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        out = torch.zeros((ctx.rows, grad.size(1)), dtype=grad.dtype, device=grad.device)
        K.scatter_rows(grad.contiguous(), idx, out)
        return out, None, None


class FramePlanner(object):
    """Builds FramePlans for one (Voxelization, DynamicVFE, SSTInputLayerV2) triple; ``supported()`` tells whether the
    configuration is one the fused kernels cover (otherwise callers use the piecewise path)."""

    def __init__(self, voxel_layer, voxel_encoder, input_layer):
        self.voxel_layer, self.vfe, self.layer = voxel_layer, voxel_encoder, input_layer
        self.grid_zyx = [int(g) for g in voxel_encoder._grid_zyx()]
        sx, sy, sz = input_layer.sparse_shape
        self.window = [int(w) for w in input_layer._window_shape3()]
        # reference_outputs no longer excludes the fused plan: those entries are formed on first access (VoxelInfo).  The
        # plan's voxel order is window-major, which the layer must allow (automatic with shuffle_voxels: see SSTInputLayerV2)
        self._ok = ([sz, sy, sx] == self.grid_zyx and input_layer.window_major
                    and self.window[0] * self.window[1] * self.window[2] <= 512)

    def supported(self, batch_size):
        cells = batch_size * self.grid_zyx[0] * self.grid_zyx[1] * self.grid_zyx[2]
        return self._ok and 1 <= batch_size <= 64 and cells <= (1 << 28)

    def build_overlapped(self, points_list, ready_event=None, points_complete=False):
        """build() on the planner's OWN stream, concurrently with whatever the caller's stream still has queued (the previous
        step's backward pass): the plan depends on the point clouds only, its ~50 short launches fit between the waves of the
        dense kernels, and its sizes are on the host long before finalize() asks for them - the host never falls behind the
        device at the head of a step (with everything on one stream it waited there for the previous step to drain, and the
        device then idled while the host woke up and issued the voxel encoder).

        The side stream waits for `ready_event` (recorded behind the producer of the point clouds, e.g. the host-to-device copy
        of a data-loader hook) and nothing else; WITHOUT one it waits for everything the caller's current stream has queued up
        to this call (an event recorded here): a cloud still being produced there - `p.cuda(non_blocking=True)`, a device-side
        augmentation - is complete before the plan reads it (a docstring promise was the only guard).
        `points_complete=True` states that the clouds need no wait at all (the round-5 behaviour).
        Memory: everything the plan keeps is allocated from the side stream's pool and marked as used by the caller's stream
        (`record_stream`), which waits for the plan before its first consumer."""
        dev = points_list[0].device
        main = torch.cuda.current_stream(dev)
        side = self.__dict__.get('_side_stream')
        if side is None or side.device != dev:
            side = self._side_stream = torch.cuda.Stream(device=dev, priority=-1)
        if ready_event is None and not points_complete:
            ready_event = torch.cuda.Event()
            ready_event.record(main)     # the producer work queued so far - not what the caller queues after this call
        if ready_event is not None:
            side.wait_event(ready_event)
        with torch.cuda.stream(side):
            plan = self.build(points_list)
            done = torch.cuda.Event()
            done.record(side)
        seen = set()

        def mark(v, depth=0):
            if isinstance(v, torch.Tensor):
                if v.is_cuda and v.untyped_storage().data_ptr() not in seen:
                    seen.add(v.untyped_storage().data_ptr())
                    v.record_stream(main)
            elif isinstance(v, (list, tuple)):
                for u in v:
                    mark(u, depth + 1)
            elif isinstance(v, dict):
                for u in v.values():
                    mark(u, depth + 1)
            elif depth < 3 and not isinstance(v, (torch.cuda.Event, torch.cuda.Stream, type, str, int, float)) and v is not None:
                for u in list(getattr(v, '__dict__', {}).values()) + [getattr(v, a, None) for a in getattr(type(v), '__slots__', ())]:
                    mark(u, depth + 1)
        mark(plan)
        main.wait_event(done)
        plan.overlapped = True
        return plan

    def build_overlapped_from_host(self, host_points, device):
        """the data loader's batch (host tensors, ideally pinned) -> device copies on the side stream -> build_overlapped: the
        copy, the plan and the size read-back all run beside the step in flight, and the copies belong to the plan
        (`plan.points_list`: what extract_feat would have been handed)"""
        device = torch.device(device)
        if device.index is None:          # 'cuda' never equals a stream's 'cuda:0': a new stream per call otherwise
            device = torch.device('cuda', torch.cuda.current_device())
        side = self.__dict__.get('_side_stream')
        if side is None or side.device != device:
            side = self._side_stream = torch.cuda.Stream(device=device, priority=-1)
        main = torch.cuda.current_stream(device)
        with torch.cuda.stream(side):
            dev_points = [p.to(device, non_blocking=True) for p in host_points]
            copied = torch.cuda.Event()
            copied.record(side)
        for p in dev_points:
            p.record_stream(main)
        plan = self.build_overlapped(dev_points, copied)
        plan.points_list = dev_points
        return plan

    @torch.no_grad()
    def build(self, points_list):
        lib = _lib.load()
        layer, vfe = self.layer, self.vfe
        layer.set_drop_info()
        bsz = len(points_list)
        points, coors = self.voxel_layer.voxelize_batch(points_list)
        n = coors.size(0)
        dev = coors.device
        gz, gy, gx = self.grid_zyx
        plan = FramePlan()
        plan.points, plan.coors, plan.n_points = points, coors, n
        plan.batch_size = bsz
        # 1. sorted-unique voxel keys of the points (count stays on the device)
        groups = K.unique_rows(coors, [0, -1, -1, -1], [bsz, gz + 1, gy + 1, gx + 1], invalid_if_negative=2,
                               defer_count=True)
        plan.groups = groups
        n_up = max(n, 1)
        plan.n_upper = n_up

        def e32(k):
            return torch.empty(k, dtype=torch.int32, device=dev)

        plan.vcoors = torch.empty((n_up, 4), dtype=torch.int32, device=dev)
        plan.gidx, plan.coors_map = e32(n_up), e32(n_up)
        plan.d_counts = e32(8)
        plan.dropped_gidx = e32(bsz)
        grid = e32(bsz * gz * gy * gx)
        rc = lib.sst_frame_voxels_i32(_lib.ptr(groups.ukeys), _lib.ptr(groups.inverse), _lib.ptr(groups.num), n, bsz, _lib.i32array(self.grid_zyx),
                                      1 if vfe.reference_compat else 0, _lib.ptr(plan.vcoors), _lib.ptr(plan.gidx),
                                      _lib.ptr(plan.coors_map), _lib.ptr(grid), _lib.ptr(plan.d_counts),
                                      _lib.ptr(plan.dropped_gidx), _lib.stream_ptr())
        _lib.check(rc, 'sst_frame_voxels_i32')
        plan.coors_map = plan.coors_map[:n]
        # 2. window bucketing, drop, window CSR of both partitions
        _, levels = layer._levels()
        flat = []
        for (cap, lo, hi) in levels:
            flat += [int(cap), int(lo), int(min(hi, 2 ** 31 - 1))]
        plan.max_tokens_cap = max(l[0] for l in levels)
        n_win = bsz * int(lib.sst_frame_windows_per_sample(_lib.i32array(self.grid_zyx), _lib.i32array(self.window)))
        plan.feat_index = torch.empty(n_up, dtype=torch.int64, device=dev)
        plan.feat_index_i32 = e32(n_up)
        plan.out_coors = torch.empty((n_up, 4), dtype=torch.int64, device=dev)
        plan.tok1, plan.posidx0, plan.posidx1 = e32(n_up), e32(n_up), e32(n_up)
        plan.winoff0, plan.winoff1 = e32(n_win + 1), e32(n_win + 1)
        seed = int(torch.randint(1, 2 ** 31 - 1, (1,)).item()) if layer.shuffle_voxels else 0   # host RNG, no sync
        ws = _lib.workspace(lib.sst_window_plan_workspace_bytes(n_up, n_win), dev)
        rc = lib.sst_window_plan_i32(_lib.ptr(plan.vcoors), _lib.ptr(grid), n_up, bsz, _lib.i32array(self.grid_zyx),
                                     _lib.i32array(self.window), _lib.i32array(flat), len(levels), seed,
                                     _lib.ptr(plan.feat_index), _lib.ptr(plan.feat_index_i32), _lib.ptr(plan.out_coors),
                                     _lib.ptr(plan.tok1),
                                     _lib.ptr(plan.winoff0), _lib.ptr(plan.winoff1), _lib.ptr(plan.posidx0),
                                     _lib.ptr(plan.posidx1), _lib.ptr(plan.d_counts), _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, 'sst_window_plan_i32')
        # the sizes travel to PINNED host memory behind the plan's kernels.  The pinned words (and the event that says they have
        # arrived) come from a free list the planner owns: a plan takes a slot here and hands it back when finalize() has read it
        # (or when the plan dies unread) - in steady state nothing is allocated.  Allocating them per plan went through the
        # pinned-memory cache, which now and then had no free block and called hipHostMalloc: a 40-55 ms host stall in ONE step
        # of a run (bench.py's reduced-precision leg: 129 instead of 160 frames/s with one such step among twenty; profiles/r05).
        # A slot belongs to ONE plan until that plan gives it up: a plan built ahead may wait arbitrarily long for its finalize()
        # (a first version recycled slots round-robin after 16 builds and handed a waiting plan the sizes of another frame).
        plan._slot_home = self._free_count_slots()
        plan.h_counts, plan.counts_ready = plan._slot_home.pop() if plan._slot_home else (
            torch.empty(8, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        plan.h_counts.copy_(plan.d_counts, non_blocking=True)
        plan.counts_ready.record()
        return plan

    def _free_count_slots(self):
        slots = self.__dict__.get('_count_slots')
        if slots is None:
            slots = self._count_slots = []
        return slots
