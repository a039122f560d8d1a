"""SSTInputLayerV2: regional grouping, voxel drop / region batching, flat<->window index precompute.

Mirror of mmdet3d/models/middle_encoders/sst_input_layer_v2.py:40-330 (same constructor kwargs, same
``forward(voxel_feats, voxel_coors, batch_size=None)``, same keys in the returned ``voxel_info`` dict).

What differs underneath: the reference runs ~100 small ATen launches with a dozen host syncs (bincount,
boolean-mask compaction x6, per-level unique/sort, padded pos/mask tensors).  Here two calls into
libsst_amd.so (sst_window_coors + sst_region_batching) produce every index on the device in int32, one
8-int readback gives the sizes, and the SRA kernels consume the resulting window CSR ("plan") directly.
The per-level padded dictionaries of the reference API (flat2win_inds / pos_dict / key_mask) are part of
the returned ``voxel_info`` when ``reference_outputs=True`` (default) so reference-style consumers keep
working, but they are formed ON FIRST ACCESS (``VoxelInfo``): the SST backbone of this package only reads
``voxel_info['sra_plan_shift{i}']`` and the positional (table, row index) pair, so a model built from a
shipped config never pays for them.

In-window order: ascending voxel index (stable); the reference's TorchEx kernel leaves it unspecified
(SURVEY.md §7 "hard parts").  With ``shuffle_voxels=True`` the drop is uniform, as in the reference.
"""
import math

import numpy as np
import torch
from torch import nn

from . import kernels as K
from .registry import MIDDLE_ENCODERS
from .sst_ops import flat2window, flat2window_v2, window2flat, window2flat_v2


class VoxelInfo(dict):
    """The ``voxel_info`` dictionary of SSTInputLayerV2 with its reference-style entries (per-voxel window ids and drop
    levels, flat2win dictionaries, padded positional tensors, key masks: sst_input_layer_v2.py:99-126) formed on first
    access.  Behaves as the plain dict the reference returns: ``in`` / ``get`` / ``[]`` see the deferred keys, and anything
    that enumerates the dictionary (``keys``, ``items``, iteration, ``len``, ``copy`` into a plain dict) forms them first."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._deferred = {}          # key -> function(self) that stores a GROUP of keys (that one among them)

    def defer(self, keys, fn):
        for k in keys:
            self._deferred[k] = fn

    def __missing__(self, key):
        fn = self._deferred.get(key)
        if fn is None:
            raise KeyError(key)
        group = [k for k, f in self._deferred.items() if f is fn]
        for k in group:                  # before the call: fn stores the group's keys through __setitem__ / update
            del self._deferred[k]
        try:
            fn(self)
        except BaseException:
            for k in group:
                self._deferred[k] = fn
            raise
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._deferred

    def get(self, key, default=None):
        return self[key] if key in self else default

    def peek(self, key, default=None):
        """the value if it is already formed (never triggers a deferred group)"""
        return dict.get(self, key, default)

    def materialize(self):
        for key in list(self._deferred):
            if key in self._deferred:
                self[key]
        return self

    def __iter__(self):
        return dict.__iter__(self.materialize())

    def keys(self):
        return dict.keys(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def values(self):
        return dict.values(self.materialize())

    def __len__(self):
        return dict.__len__(self) + len(self._deferred)

    def shallow(self):
        """a copy that shares the values and keeps the deferred groups deferred (they store into the copy they are asked from)"""
        out = VoxelInfo()
        dict.update(out, dict.items(self))
        out._deferred = dict(self._deferred)
        return out

    # the rest of the dict protocol on deferred keys: a write or a removal settles the key first, copies get
    # their own table of deferred groups
    def __copy__(self):
        return self.shallow()

    def copy(self):
        return self.shallow()

    def __reduce__(self):                     # pickling / deepcopy: the formed dictionary
        return (dict, (dict(dict.items(self.materialize())),))

    def _drop_deferred(self, key):
        """``key`` is about to be overwritten or removed: form its group first (the other keys of the group stay valid), so that
        no stale deferred entry is left to overwrite the user's value or to be counted twice"""
        if key in self._deferred:
            self[key]

    def __setitem__(self, key, value):
        self._drop_deferred(key)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._drop_deferred(key)
        dict.__delitem__(self, key)

    def pop(self, key, *default):
        self._drop_deferred(key)
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        dict.__setitem__(self, key, default)
        return default

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v


@MIDDLE_ENCODERS.register_module()
class PseudoMiddleEncoderForSpconvFSD(nn.Module):
    """sst_input_layer_v2.py:15-37: identity wrapper used by the spconv FSD configs."""

    def __init__(self, ):
        super().__init__()

    def forward(self, voxel_feats, voxel_coors, batch_size=None):
        return {'voxel_feats': voxel_feats, 'voxel_coors': voxel_coors}


def _nonzero_known(mask, count):
    """indices of the set entries of a 1-D mask whose number is already known on the host: no readback"""
    if hasattr(torch, 'nonzero_static'):
        return torch.nonzero_static(mask, size=int(count)).squeeze(1)
    return torch.nonzero(mask).squeeze(1)


@MIDDLE_ENCODERS.register_module()
class SSTInputLayerV2(nn.Module):

    def __init__(self,
                 drop_info,
                 window_shape,
                 sparse_shape,
                 shuffle_voxels=True,
                 debug=True,
                 normalize_pos=False,
                 pos_temperature=10000,
                 mute=False,
                 reference_outputs=True,
                 window_major=None,
                 ):
        super().__init__()
        self.fp16_enabled = False
        self.meta_drop_info = drop_info
        self.sparse_shape = sparse_shape
        self.shuffle_voxels = shuffle_voxels
        self.debug = debug
        self.window_shape = window_shape
        self.normalize_pos = normalize_pos
        self.pos_temperature = pos_temperature
        self.mute = mute
        self.reference_outputs = reference_outputs
        # window_major: emit the kept voxels ordered by their regular (shift-0) window instead of the (shuffled)
        # input order.  The order of the voxel list carries no meaning downstream (features travel with their
        # coordinates), but the attention kernels then read every window as one contiguous run of rows instead of
        # 30-100 scattered ones.  Costs nothing: the permutation is folded into the gather that removes the dropped
        # voxels, and every per-voxel entry of voxel_info (reference-style ones included) is emitted in that order.
        # None (default) = automatic: on when shuffle_voxels is on - the reference's own output order is then a
        # uniformly random permutation (sst_input_layer_v2.py:93-97), i.e. no consumer can depend on it - and off
        # without the shuffle, where the reference keeps the input order and so does this layer.
        self.window_major = bool(shuffle_voxels) if window_major is None else bool(window_major)

    # ---------------------------------------------------------------------------------------
    def set_drop_info(self):
        if hasattr(self, 'drop_info'):
            return
        meta = self.meta_drop_info
        if isinstance(meta, tuple):
            self.drop_info = meta[0] if self.training else meta[1]
        else:
            self.drop_info = meta
        if not self.mute:
            print(f'drop_info is set to {self.drop_info}, in input_layer')

    def _window_shape3(self):
        ws = tuple(self.window_shape)
        if len(ws) == 2:
            return (ws[0], ws[1], self.sparse_shape[-1])
        return ws

    def _levels(self):
        keys = list(self.drop_info.keys())
        levels = []
        for dl in keys:
            lo, hi = self.drop_info[dl]['drop_range']
            levels.append((self.drop_info[dl]['max_tokens'], lo, hi))
        return keys, levels

    # ---------------------------------------------------------------------------------------
    def forward(self, voxel_feats, voxel_coors, batch_size=None):
        '''
        Args:
            voxel_feats: shape=[N, C], N is the voxel num in the batch.
            voxel_coors: shape=[N, 4], [b, z, y, x]
        Returns:
            voxel_info: dict, same keys as the reference (sst_input_layer_v2.py:99-126) plus
                        'sra_plan_shift{i}', 'pos_embed_shift{i}'.
        '''
        from . import _lib
        voxel_feats = _lib.as_fp32(voxel_feats)    # auto_fp16 of the reference names a non-existent argument (sst_input_layer_v2.py:79): fp32 passes
        return self.apply_plan(self.build_plan(voxel_coors, batch_size, voxel_feats.size(1), voxel_feats.dtype),
                               voxel_feats)

    @torch.no_grad()
    def build_plan(self, voxel_coors, batch_size=None, feat_dim=128, dtype=torch.float32):
        """Everything of forward() that depends only on the voxel coordinates (window bucketing, drop / shuffle,
        window CSR, positional embeddings): no voxel features, no parameters.  A caller that pipelines frames can
        run it ahead of time (another stream, the data loader) and finish with apply_plan(plan, voxel_feats)."""
        self.set_drop_info()
        if voxel_coors.dim() != 2 or voxel_coors.size(1) != 4:
            raise RuntimeError('voxel_coors must be [N,4] (b,z,y,x)')
        voxel_coors = voxel_coors.long()

        shuffle_inds = None
        if self.shuffle_voxels:
            # random permutation = order of 24-bit random keys under the library's radix sort (3 passes; torch.randperm
            # runs a ~25-launch merge sort).  Ties (probability ~M / 2^24 per voxel) keep their input order.
            m0 = len(voxel_coors)
            if m0 > 0 and voxel_coors.is_cuda:
                keys = torch.randint(0, 1 << 24, (m0,), device=voxel_coors.device, dtype=torch.int64)
                shuffle_inds = K.sort_pairs_u64(keys, 24)[1].long()
            else:
                shuffle_inds = torch.randperm(m0, device=voxel_coors.device)
            voxel_coors = voxel_coors[shuffle_inds]
        voxel_coors = voxel_coors.contiguous()

        m = voxel_coors.size(0)
        sx, sy, sz = self.sparse_shape
        wx, wy, wz = self._window_shape3()
        assert sz < sx, 'Usually holds... in case of wrong order'
        level_keys, levels = self._levels()

        win0, ciw0, win1, ciw1 = K.window_coors(voxel_coors, [sx, sy, sz], [wx, wy, wz])
        per_sample = (math.ceil(sx / wx) + 1) * (math.ceil(sy / wy) + 1) * (math.ceil(sz / wz) + 1)
        if batch_size is None:
            batch_size = int(voxel_coors[:, 0].max().item()) + 1 if m > 0 else 1
        win_bits = max(1, int(per_sample * int(batch_size)).bit_length())
        rb = K.region_batching(win0, win1, win_bits, levels)
        sizes = rb['counts']
        if self.debug and m > 0:    # smallest drop level among the kept voxels (-1: a population no drop_range covers)
            lvl_min = torch.where(rb['keep'] > 0, torch.minimum(rb['level0'], rb['level1']), 0).min().reshape(1)
            sizes = torch.cat([sizes, lvl_min.to(sizes.dtype)])
        counts = sizes.tolist() + [0]  # the single readback: M', W0, W1, T0, T1, (spare), [8] = the level check
        m_keep, n_win = counts[0], (counts[1], counts[2])
        # upper bound on the tokens per window handed to the attention kernels: the largest surviving window of the
        # shift (read back with the other sizes) - lets them pick the smallest register / LDS class that fits
        max_tokens_cap = max(l[0] for l in levels)
        win_max = tuple(min(max_tokens_cap, max(1, int(t))) for t in (counts[3], counts[4]))

        voxel_info = VoxelInfo()
        keep_all = (m_keep == m)
        tok = [rb['tok0'], rb['tok1']]
        if self.window_major and m_keep > 0:
            perm = rb['tok0'][:m_keep].long()                      # kept-voxel index at every shift-0 window slot
            keep_idx = perm if keep_all else _nonzero_known(rb['keep'], m_keep).index_select(0, perm)
            inv = torch.empty(m_keep, dtype=torch.int32, device=perm.device)
            inv[perm] = torch.arange(m_keep, dtype=torch.int32, device=perm.device)
            tok = [torch.arange(m_keep, dtype=torch.int32, device=perm.device), inv[rb['tok1'][:m_keep].long()]]

            def sel(t):
                return t.index_select(0, keep_idx)
        elif keep_all:
            keep_idx = torch.arange(m, device=voxel_coors.device, dtype=torch.long)

            def sel(t):
                return t
        else:
            keep_idx = _nonzero_known(rb['keep'], m_keep)

            def sel(t):
                return t.index_select(0, keep_idx)

        voxel_coors = sel(voxel_coors)
        ciws = (sel(ciw0), sel(ciw1))

        # rows of the caller's voxel_feats that survive, in output order (shuffle and drop / re-order folded
        # into one gather); None = all rows in their own order
        if shuffle_inds is not None:
            voxel_info['_feat_index'] = shuffle_inds.index_select(0, keep_idx)
        elif keep_all and not (self.window_major and m_keep > 0):
            voxel_info['_feat_index'] = None
        else:
            voxel_info['_feat_index'] = keep_idx
        voxel_info['voxel_coors'] = voxel_coors
        voxel_info['voxel_keep_inds'] = keep_idx
        rows_in_window_order = bool(self.window_major and m_keep > 0)
        for i in range(2):
            voxel_info[f'coors_in_win_shift{i}'] = ciws[i].long()
            voxel_info[f'sra_plan_shift{i}'] = K.WindowPlan(tok[i], rb[f'winoff{i}'], n_win[i], m_keep, win_max[i],
                                                            rows_in_window_order=(rows_in_window_order and i == 0))
            voxel_info[f'pos_index_shift{i}'] = self.pos_table_index(ciws[i])
        voxel_info['pos_table'] = self.pos_table_cached(feat_dim, dtype, voxel_coors.device)
        voxel_info['batch_size'] = int(batch_size)
        # the [M, C] positional tensors: only the per-layer path adds them to x as tensors (the encoder chains take the
        # (table, row index) pair above), so they are gathered when somebody asks
        voxel_info.defer([f'pos_embed_shift{i}' for i in range(2)], lambda info: info.update(
            {f'pos_embed_shift{i}': info['pos_table'].index_select(0, info[f'pos_index_shift{i}'].long()) for i in range(2)}))

        # the per-voxel window ids / drop levels / flat2win indices, the padded positional tensors and the key masks are what
        # reference-style consumers read; the SRA kernels only need the window CSR and the positional rows: deferred
        if self.debug:
            # "a window population matched no drop_range" (the reference's range assertions, sst_input_layer_v2.py:182-187):
            # the smallest level of a kept voxel rode along with the sizes above - no extra host synchronisation
            assert counts[8] >= 0, 'a window population matched no drop_range'
        if self.reference_outputs or self.debug:
            wins_all, lvls_all, f2ws_all = (win0, win1), (rb['level0'], rb['level1']), (rb['flat2win0'], rb['flat2win1'])
            level_map = None
            if list(level_keys) != list(range(len(level_keys))):   # drop_info keyed by something else than 0..n-1
                level_map = torch.tensor(level_keys, device=voxel_coors.device, dtype=torch.long)
            self._defer_reference_entries(voxel_info, sel, wins_all, lvls_all, f2ws_all, level_keys, level_map)

        if self.shuffle_voxels:
            voxel_info['shuffle_inds'] = shuffle_inds
        return voxel_info

    _REFERENCE_KEYS = ('batch_win_inds', 'voxel_drop_level', 'flat2win_inds', 'pos_dict', 'key_mask')

    def _reference_entries(self, info, sel, wins_all, lvls_all, f2ws_all, level_keys, level_map):
        """The reference-style entries of ``voxel_info`` (sst_input_layer_v2.py:99-126): batch_win_inds / voxel_drop_level /
        flat2win_inds / pos_dict / key_mask of both shifts.  ``sel`` maps a per-input-voxel tensor to the kept voxels in
        output order."""
        lvls = tuple(sel(t) for t in lvls_all)
        f2ws = tuple(sel(t) for t in f2ws_all)
        out = {}
        for i in range(2):
            out[f'batch_win_inds_shift{i}'] = sel(wins_all[i]).long()
            lv = lvls[i].long()
            if level_map is not None:
                lv = level_map[lv.clamp(min=0)]
            out[f'voxel_drop_level_shift{i}'] = lv
            inds_dict = {}
            if lvls[i].numel() > 0:
                present = torch.stack([(lvls[i] == li).any() for li in range(len(level_keys))]).tolist()   # one read-back
            else:
                present = [False] * len(level_keys)
            for li, dl in enumerate(level_keys):
                if not present[li]:
                    continue
                mask = lvls[i] == li
                inds_dict[dl] = (f2ws[i][mask].long(), torch.where(mask))
            inds_dict['voxel_drop_level'] = lv
            inds_dict['batching_info'] = self.drop_info
            out[f'flat2win_inds_shift{i}'] = inds_dict
            out[f'pos_dict_shift{i}'] = flat2window_v2(info[f'pos_embed_shift{i}'], inds_dict)
            out[f'key_mask_shift{i}'] = self.get_key_padding_mask(inds_dict)
        if self.debug:    # the reference's round-trip check of the index dictionaries (sst_input_layer_v2.py:119-123)
            coors = info['voxel_coors']
            coors_3d_dict_shift0 = flat2window_v2(coors, out['flat2win_inds_shift0'])
            coors_2d = window2flat_v2(coors_3d_dict_shift0, out['flat2win_inds_shift0'])
            assert (coors_2d == coors).all()
        return out

    def _defer_reference_entries(self, voxel_info, sel, wins_all, lvls_all, f2ws_all, level_keys, level_map):
        """register them as ONE deferred group of ``voxel_info``, formed from the tensors region batching left"""
        def form(info):
            dict.update(info, self._reference_entries(info, sel, wins_all, lvls_all, f2ws_all, level_keys, level_map))

        voxel_info.defer([f'{name}_shift{i}' for i in range(2) for name in self._REFERENCE_KEYS], form)

    def defer_reference_entries_of_kept(self, voxel_info):
        """The same group for a plan whose kernels keep no per-voxel window ids / levels (csrc/frame_plan.hip), re-derived on
        first access from the KEPT voxels: their window ids, and region batching run on them again.  Nothing is dropped the
        second time (a window's surviving population never exceeds the max_tokens of the level that population selects), so
        the dictionaries describe exactly the windows the SRA plan holds; a window whose population crossed a drop_range
        boundary when the other shift's drop thinned it sits in the (smaller) level of its surviving population - which
        padded batch a window rides in does not enter any result.  Also forms ``coors_in_win_shift{i}``."""
        def form(info):
            self.set_drop_info()
            coors = info['voxel_coors'].contiguous()
            m = coors.size(0)
            sx, sy, sz = self.sparse_shape
            wx, wy, wz = self._window_shape3()
            level_keys, levels = self._levels()
            win0, ciw0, win1, ciw1 = K.window_coors(coors, [sx, sy, sz], [wx, wy, wz])
            per_sample = (math.ceil(sx / wx) + 1) * (math.ceil(sy / wy) + 1) * (math.ceil(sz / wz) + 1)
            win_bits = max(1, int(per_sample * int(info['batch_size'])).bit_length())
            rb = K.region_batching(win0, win1, win_bits, levels)
            if self.debug:
                assert int(rb['counts'][0].item()) == m, 'a kept voxel did not survive its own window'
            level_map = None
            if list(level_keys) != list(range(len(level_keys))):
                level_map = torch.tensor(level_keys, device=coors.device, dtype=torch.long)
            dict.update(info, {'coors_in_win_shift0': ciw0.long(), 'coors_in_win_shift1': ciw1.long()})
            dict.update(info, self._reference_entries(info, lambda t: t, (win0, win1), (rb['level0'], rb['level1']),
                                                      (rb['flat2win0'], rb['flat2win1']), level_keys, level_map))

        voxel_info.defer([f'{name}_shift{i}' for i in range(2) for name in self._REFERENCE_KEYS + ('coors_in_win',)], form)

    def apply_plan(self, plan, voxel_feats):
        """voxel_info of forward(): the plan plus the surviving voxel features in plan order (one gather)."""
        voxel_info = plan.shallow() if isinstance(plan, VoxelInfo) else dict(plan)
        idx = voxel_info.pop('_feat_index')
        voxel_info['voxel_feats'] = voxel_feats if idx is None else voxel_feats.index_select(0, idx)
        return voxel_info

    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def pos_table(self, feat_dim, dtype, device):
        """All distinct positional embeddings: one row per in-window coordinate, row index
        (z * wy + y) * wx + x.  Same arithmetic as get_pos_embed (sst_input_layer_v2.py:238-305)."""
        wx, wy, wz3 = self._window_shape3()
        window_shape = self.window_shape
        if len(window_shape) == 2 or window_shape[-1] == 1:
            ndim = 2
            win_x, win_y = window_shape[:2]
            win_z = 0
            nz = wz3
        else:
            win_x, win_y, win_z = window_shape
            ndim = 3
            nz = win_z
        zz, yy, xx = torch.meshgrid(torch.arange(nz, device=device), torch.arange(wy, device=device),
                                    torch.arange(wx, device=device), indexing='ij')
        z = zz.reshape(-1) - win_z / 2
        y = yy.reshape(-1) - win_y / 2
        x = xx.reshape(-1) - win_x / 2
        if self.normalize_pos:
            x = x / win_x * 2 * 3.1415  # [-pi, pi]
            y = y / win_y * 2 * 3.1415
            z = z / win_z * 2 * 3.1415
        pos_length = feat_dim // ndim
        inv_freq = torch.arange(pos_length, dtype=torch.float32, device=device)
        inv_freq = self.pos_temperature ** (2 * (inv_freq // 2) / pos_length)
        embed_x = x[:, None] / inv_freq[None, :]
        embed_y = y[:, None] / inv_freq[None, :]
        embed_x = torch.stack([embed_x[:, ::2].sin(), embed_x[:, 1::2].cos()], dim=-1).flatten(1)
        embed_y = torch.stack([embed_y[:, ::2].sin(), embed_y[:, 1::2].cos()], dim=-1).flatten(1)
        if ndim == 3:
            embed_z = z[:, None] / inv_freq[None, :]
            embed_z = torch.stack([embed_z[:, ::2].sin(), embed_z[:, 1::2].cos()], dim=-1).flatten(1)
            pos = torch.cat([embed_x, embed_y, embed_z], dim=-1).to(dtype)
        else:
            pos = torch.cat([embed_x, embed_y], dim=-1).to(dtype)
        gap = feat_dim - pos.size(1)
        assert gap >= 0
        if gap > 0:
            assert ndim == 3
            pos = torch.cat([pos, torch.zeros((pos.size(0), gap), dtype=dtype, device=device)], dim=1)
        return pos

    def pos_table_cached(self, feat_dim, dtype, device):
        """pos_table of the current configuration, built once (it only depends on the configuration)"""
        key = (feat_dim, dtype, str(device), self.pos_temperature, self.normalize_pos)
        cache = self.__dict__.setdefault('_pos_table_cache', {})
        table = cache.get(key)
        if table is None:
            table = cache[key] = self.pos_table(feat_dim, dtype, device)
        return table

    @torch.no_grad()
    def get_pos_embed_flat(self, coors_in_win, feat_dim, dtype):
        """[M, feat_dim] positional embedding of every voxel (flat layout)."""
        wx, wy, _ = self._window_shape3()
        table = self.pos_table_cached(feat_dim, dtype, coors_in_win.device)
        return table.index_select(0, self.pos_table_index(coors_in_win).long())

    def pos_table_index(self, coors_in_win):
        """row of pos_table for in-window coordinates (z, y, x): (z * wy + y) * wx + x, int32"""
        wx, wy, _ = self._window_shape3()
        c = coors_in_win
        return ((c[:, 0] * wy + c[:, 1]) * wx + c[:, 2]).to(torch.int32)

    @torch.no_grad()
    def get_pos_embed(self, inds_dict, coors_in_win, feat_dim, dtype):
        """Reference signature: per-level padded dict of positional embeddings."""
        return flat2window_v2(self.get_pos_embed_flat(coors_in_win, feat_dim, dtype), inds_dict)

    @torch.no_grad()
    def get_key_padding_mask(self, ind_dict):
        num_all_voxel = len(ind_dict['voxel_drop_level'])
        key_padding = torch.ones((num_all_voxel, 1), device=ind_dict['voxel_drop_level'].device).bool()
        window_key_padding_dict = flat2window_v2(key_padding, ind_dict)
        for key, value in window_key_padding_dict.items():  # True = padded slot
            window_key_padding_dict[key] = value.logical_not().squeeze(2)
        return window_key_padding_dict


@MIDDLE_ENCODERS.register_module()
class SSTInputLayer(nn.Module):
    """First-generation input layer (mmdet3d/models/middle_encoders/sst_input_layer.py:14-364): 2-D windows,
    ``shifts_list`` instead of a fixed half-window shift, returns ``(voxel_feat, flat2win_inds_list, voxel_info)``
    and leaves positional embedding / key masks to the SSTv1 backbone.  Same constructor kwargs and outputs;
    the grouping, drop and index computation run on the same HIP kernels as SSTInputLayerV2 (window ids follow
    the v1 numbering of sst_input_layer.py:299-330).  Accepts the optional third positional argument that
    DynamicVoxelNet passes (detectors/dynamic_voxelnet.py:43)."""

    def __init__(self, drop_info, shifts_list, window_shape, point_cloud_range, voxel_size, shuffle_voxels=True,
                 debug=True):
        super().__init__()
        self.fp16_enabled = False
        self.meta_drop_info = drop_info
        self.shifts_list = shifts_list
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.shuffle_voxels = shuffle_voxels
        self.debug = debug
        self.window_shape = window_shape

    def set_drop_info(self):
        if hasattr(self, 'drop_info'):
            return
        meta = self.meta_drop_info
        if isinstance(meta, tuple):
            self.drop_info = meta[0] if self.training else meta[1]
        else:
            self.drop_info = meta

    @torch.no_grad()
    def window_partition(self, coors, voxel_info):
        """v1 window ids / in-window coordinates, exactly sst_input_layer.py:299-330."""
        win_shape_x, win_shape_y = self.window_shape
        pc_range, voxel_size = self.point_cloud_range, self.voxel_size
        bev_shape_x = int(np.ceil((pc_range[3] - pc_range[0]) / voxel_size[0]))
        bev_shape_y = int(np.ceil((pc_range[4] - pc_range[1]) / voxel_size[1]))
        max_num_win_x = int(np.ceil((bev_shape_x / win_shape_x)) + 1)
        max_num_win_y = int(np.ceil((bev_shape_y / win_shape_y)) + 1)
        max_num_win_per_sample = max_num_win_x * max_num_win_y
        for i, (shift_x, shift_y) in enumerate(self.shifts_list):
            assert shift_x == 0 or shift_x == win_shape_x // 2, 'Usually ...'
            sx = coors[:, 3] + (win_shape_x - shift_x if shift_x > 0 else 0)
            sy = coors[:, 2] + (win_shape_y - shift_y if shift_y > 0 else 0)
            wxi = torch.div(sx, win_shape_x, rounding_mode='floor')
            wyi = torch.div(sy, win_shape_y, rounding_mode='floor')
            voxel_info[f'batch_win_inds_shift{i}'] = coors[:, 0] * max_num_win_per_sample + wxi * max_num_win_y + wyi
            voxel_info[f'coors_in_win_shift{i}'] = torch.stack([sx - wxi * win_shape_x, sy - wyi * win_shape_y], dim=-1)
        self._win_id_bound = max_num_win_per_sample
        return voxel_info

    def forward(self, voxel_feat, coors, batch_size=None):
        self.set_drop_info()
        if len(self.shifts_list) != 2:
            raise NotImplementedError('SSTInputLayer: two shift layouts are expected (every shipped config)')
        coors = coors.long()
        if self.shuffle_voxels:
            shuffle_inds = torch.randperm(len(voxel_feat), device=voxel_feat.device)
            voxel_feat = voxel_feat[shuffle_inds]
            coors = coors[shuffle_inds]
        voxel_info = self.window_partition(coors, {})
        m = coors.size(0)
        keys = list(self.drop_info.keys())
        levels = [(self.drop_info[k]['max_tokens'],) + tuple(self.drop_info[k]['drop_range']) for k in keys]
        with torch.no_grad():
            if batch_size is None:
                batch_size = int(coors[:, 0].max().item()) + 1 if m > 0 else 1
            win_bits = max(1, int(self._win_id_bound * int(batch_size)).bit_length())
            w0 = voxel_info['batch_win_inds_shift0'].int().contiguous()
            w1 = voxel_info['batch_win_inds_shift1'].int().contiguous()
            rb = K.region_batching(w0, w1, win_bits, levels)
            counts = rb['counts'].tolist()
        m_keep = counts[0]
        if m_keep == m:
            keep_idx = torch.arange(m, device=coors.device, dtype=torch.long)
        else:
            keep_idx = _nonzero_known(rb['keep'], m_keep)
        sel = (lambda t: t) if m_keep == m else (lambda t: t.index_select(0, keep_idx))
        voxel_feat = sel(voxel_feat)
        coors = sel(coors)
        key_map = torch.tensor(keys, device=coors.device, dtype=torch.long)
        out_info = {'voxel_keep_inds': keep_idx, 'coors': coors}
        flat2win_inds_list = []
        cap = max(l[0] for l in levels)
        for i in range(2):
            out_info[f'batch_win_inds_shift{i}'] = sel(voxel_info[f'batch_win_inds_shift{i}'])
            out_info[f'coors_in_win_shift{i}'] = sel(voxel_info[f'coors_in_win_shift{i}'])
            lvl_idx = sel(rb[f'level{i}'])
            if self.debug:
                assert (lvl_idx >= 0).all()
            out_info[f'voxel_drop_level_shift{i}'] = key_map[lvl_idx.long().clamp(min=0)]
            f2w = sel(rb[f'flat2win{i}'])
            inds = {}
            for li, dl in enumerate(keys):
                mask = lvl_idx == li
                if mask.any():
                    inds[dl] = (f2w[mask].long(), torch.where(mask))
            flat2win_inds_list.append(inds)
            out_info[f'sra_plan_shift{i}'] = K.WindowPlan(rb[f'tok{i}'], rb[f'winoff{i}'], counts[1 + i], m_keep, cap)
        if self.debug:
            c3d = flat2window(coors, out_info['voxel_drop_level_shift0'], flat2win_inds_list[0], self.drop_info)
            assert (window2flat(c3d, flat2win_inds_list[0]) == coors).all()
        return voxel_feat, flat2win_inds_list, out_info
