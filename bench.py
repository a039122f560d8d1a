#!/usr/bin/env python
"""bench.py — LiDAR frames/s of the SST backbone hot path (fwd+bwd) at Waymo 0.32 m voxels on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames resident in HBM:
    dynamic voxelize -> DynamicVFE (3 segmented scatters) -> SSTInputLayerV2 (window bucketing / region
    batching) -> 6 SRA blocks (12 encoder layers), forward + backward; with --gpus N > 1 the gradients are
    all-reduced over RCCL (one flat bucket) inside the step.
Workload at N=1 (BASELINE.json configs[1]): uniform synthetic cloud, 116 000 points/frame ->
~90 k non-empty voxels, 1 frame per GPU; frames shard data-parallel across ranks (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel group = the SRA attention core forward (sra_fwd_mfma_k<NT> launch group):
                achieved = 2056 B/token x tokens / mean HIP-event duration of the group over the timed region,
                vs the 8 TB/s HBM peak.
  cpu_baseline  the CPU port of the reference path (oracle/cpu_pipeline.py) timed on the host cores for one
                frame of the same workload, fwd+bwd (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

VOXEL_SIZE = (0.32, 0.32, 6)
PC_RANGE = [-74.88, -74.88, -2, 74.88, 74.88, 4]
DROP_TRAIN = {0: {'max_tokens': 30, 'drop_range': (0, 30)}, 1: {'max_tokens': 60, 'drop_range': (30, 60)},
              2: {'max_tokens': 100, 'drop_range': (60, 100000)}}
DROP_TEST = {0: {'max_tokens': 30, 'drop_range': (0, 30)}, 1: {'max_tokens': 60, 'drop_range': (30, 60)},
             2: {'max_tokens': 100, 'drop_range': (60, 100)}, 3: {'max_tokens': 144, 'drop_range': (100, 100000)}}
HBM_PEAK_GBS = 8000.0
SRA_BYTES_PER_TOKEN = 4 * 128 * 4 + 8   # Q,K,V read + O write (fp32, d=128) + index: SURVEY.md §8d(5)
SRA_BWD_BYTES_PER_TOKEN = 8 * 128 * 4 + 8   # Q,K,V,O,dO read + dQ,dK,dV written + index


def make_cloud(n, seed, device, channels=3):
    """SURVEY.md section 8(d) U-cloud: uniform in the range; channels beyond xyz (intensity, elongation ...) uniform in [0, 1)"""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([149.76, 149.76, 6.0]) + torch.tensor([-74.88, -74.88, -2.0])
    if channels > 3:
        xyz = torch.cat([xyz, torch.rand(n, channels - 3, generator=g)], 1)
    return xyz.to(device)


def load_config_fixture(name):
    """resolved `model` dictionary of a shipped config (tests/golden/configs/<name>.model.py, checked against the config text
    by tests/test_config_fixtures.py)"""
    import ast
    return ast.literal_eval(open(os.path.join(ROOT, 'tests', 'golden', 'configs', name + '.model.py')).read())


def make_lidar_cloud(seed, device, beams=64, azimuth_steps=2650, sensor_height=2.0, outside_fraction=0.05,
                     duplicate_fraction=0.02):
    """SURVEY.md section 8(d) "L-cloud" + its pathological input in one frame: a spinning 64-beam sensor 2 m above a ground
    plane (range = ray / ground intersection, 2 cm noise; rays above the horizon hit a far wall at 75 m), 5 % of the rays
    replaced by box surfaces, then 5 % of the points pushed OUTSIDE the point-cloud range (the voxelizer's clamp) and 2 %
    duplicated exactly.  ~170 k points that pile up near the sensor: few, crowded windows - the opposite of the uniform cloud."""
    g = torch.Generator().manual_seed(seed)
    az = torch.arange(azimuth_steps, dtype=torch.float32) * (2 * 3.14159265 / azimuth_steps)
    el = torch.linspace(-25.0, 3.0, beams) * (3.14159265 / 180)
    el, az = torch.meshgrid(el, az, indexing='ij')
    el, az = el.reshape(-1), az.reshape(-1)
    rng = torch.where(el < -0.01, sensor_height / torch.tan(-el).clamp(min=1e-3), torch.full_like(el, 75.0))
    rng = (rng + 0.02 * torch.randn(rng.shape, generator=g)).clamp(1.0, 75.0)
    xyz = torch.stack([rng * torch.cos(el) * torch.cos(az), rng * torch.cos(el) * torch.sin(az),
                       sensor_height - 2.0 + rng * torch.sin(el) - 0.0], 1)
    xyz[:, 2] -= 1.8                                              # ground near z = -1.8 (the range is z in [-2, 4])
    n = xyz.size(0)
    box = torch.rand(n, generator=g) < 0.05                       # rays that hit an object instead
    centres = (torch.rand(40, 3, generator=g) - 0.5) * torch.tensor([100.0, 100.0, 0.0]) + torch.tensor([0.0, 0.0, -0.9])
    which = torch.randint(0, 40, (n,), generator=g)
    on_box = centres[which] + (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([4.2, 1.9, 1.7])
    xyz = torch.where(box[:, None], on_box, xyz)
    out = torch.rand(n, generator=g) < outside_fraction            # outside the range: clamped by the voxelizer
    xyz = torch.where(out[:, None], xyz * torch.tensor([1.0, 1.0, 1.0]) + torch.sign(xyz) * torch.tensor([80.0, 80.0, 0.0]), xyz)
    dup = torch.randint(0, n, (int(n * duplicate_fraction),), generator=g)
    xyz = torch.cat([xyz, xyz[dup]])
    return xyz[torch.randperm(xyz.size(0), generator=g)].contiguous().to(device)


def _pipeline_config(num_blocks=6, with_bev=False):
    """the `model` dictionary of the headline: configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:26-79 with the debug
    checks and prints off, and without the dense BEV neck / head unless `with_bev`"""
    return dict(
        type='DynamicVoxelNet',
        voxel_layer=dict(voxel_size=VOXEL_SIZE, max_num_points=-1, point_cloud_range=PC_RANGE, max_voxels=(-1, -1)),
        voxel_encoder=dict(
            type='DynamicVFE', in_channels=3, feat_channels=[64, 128], with_distance=False, voxel_size=VOXEL_SIZE,
            with_cluster_center=True, with_voxel_center=True, point_cloud_range=PC_RANGE,
            norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)),
        middle_encoder=dict(
            type='SSTInputLayerV2', window_shape=(12, 12, 1), sparse_shape=(468, 468, 1), shuffle_voxels=True,
            window_major=True,
            debug=False, drop_info=(DROP_TRAIN, DROP_TEST), pos_temperature=10000, normalize_pos=False, mute=True,
            reference_outputs=False),
        backbone=dict(
            type='SSTv2', d_model=[128] * num_blocks, nhead=[8] * num_blocks, num_blocks=num_blocks,
            dim_feedforward=[256] * num_blocks, output_shape=[468, 468], debug=False,
            # --workload sst_bev: the backbone as the config builds it (sst_waymoD5_1x_3class_8heads_v2.py:64-79): BEV canvas +
            # three dilated 3 x 3 convolutions (MIOpen); the headline leaves them out
            **(dict(num_attached_conv=3, to_bev=True, conv_in_channel=128, conv_out_channel=128,
                    conv_kwargs=[dict(kernel_size=3, dilation=1, padding=1, stride=1),
                                 dict(kernel_size=3, dilation=1, padding=1, stride=1),
                                 dict(kernel_size=3, dilation=2, padding=2, stride=1)])
               if with_bev else dict(num_attached_conv=0, to_bev=False))))


def Pipeline(num_blocks=6, with_bev=False, model_cfg=None, voxel_feats_only=None):
    """The SST-base hot path as the detector that calls it: sst_amd.DynamicVoxelNet (mirror of detectors/dynamic_voxelnet.py:10-71)
    built through the registry from a config `model` dictionary - the headline's (_pipeline_config) or, `model_cfg`, any other
    (the shipped configs read from tests/golden/configs/).  forward(points_list, prepared=None) -> features of the kept voxels
    [M', C] (`voxel_feats_only`, default: whenever the backbone does not go to the BEV canvas ... or is asked not to), or the
    backbone's own output (BEV canvas behind the attached convolutions)."""
    import sst_amd

    class _Pipeline(sst_amd.DynamicVoxelNet):

        def forward(self, points_list, prepared=None):
            info = self.voxel_info(points_list, prepared)
            self.last_voxel_coors = info['voxel_coors']
            self.last_plans = [info.get('sra_plan_shift0'), info.get('sra_plan_shift1')]
            if self.voxel_feats_only:
                return self.backbone.forward_voxels(info)
            out = self.backbone(info)[0]
            return out if self.backbone.to_bev else out['voxel_feats']

    cfg = dict(model_cfg if model_cfg is not None else _pipeline_config(num_blocks, with_bev))
    cfg.pop('type', None)
    model = _Pipeline(**cfg)
    model.with_bev = bool(model.backbone.to_bev)
    model.voxel_feats_only = (not model.with_bev) if voxel_feats_only is None else bool(voxel_feats_only)
    return model


class GcWatch(object):
    """Python's cyclic collector under watch: every collection (generation, duration) with its end time, so that a timed
    region can say how much of it was spent inside the collector (a full collection of a torch process walks ~10^6 objects:
    tens of milliseconds - one outlier step among twenty; VERDICT round 4 item 3 asked which it was)."""

    def __init__(self):
        import gc
        self.events, self._t0 = [], None
        gc.callbacks.append(self._cb)

    def _cb(self, phase, info):
        if phase == 'start':
            self._t0 = time.perf_counter()
        elif self._t0 is not None:
            now = time.perf_counter()
            self.events.append((now, int(info.get('generation', -1)), now - self._t0))

    def window(self, t_begin, t_end):
        sel = [e for e in self.events if t_begin <= e[0] <= t_end + 1e-3]
        return {'collections': len(sel), 'full_collections': sum(1 for e in sel if e[1] == 2),
                'ms_total': round(1e3 * sum(e[2] for e in sel), 3), 'ms_longest': round(1e3 * max([e[2] for e in sel] or [0.0]), 3)}


GC_WATCH = None


def note(msg):
    """progress marker on stderr (the line itself goes to stdout): which leg a run was in when something went wrong"""
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


class StepTimes(object):
    """Per-step durations of a timed loop WITHOUT synchronising inside it: one event recorded on the stream behind every step,
    the differences read after the loop's final synchronisation (device-side time between the ends of consecutive steps).  The
    line carries median / min / max of every leg: a mean alone hides an outlier step (bf16 leg)."""

    def __init__(self):
        self.events, self.host, self.allocs = [], [], []
        self.t_begin = time.perf_counter()
        self.allocs0 = torch.cuda.memory_stats().get('num_device_alloc', 0) if torch.cuda.is_available() else 0

    def close(self):
        """end of the timed loop (call right behind its final synchronisation): what stats() reports is what happened up to here"""
        if getattr(self, 't_end', None) is None:
            self.t_end = time.perf_counter()
            self.allocs1 = torch.cuda.memory_stats().get('num_device_alloc', 0) if torch.cuda.is_available() else 0
        return self

    def gc(self):
        """the collector's share of the loop this object timed"""
        self.close()
        return GC_WATCH.window(self.t_begin, self.t_end) if GC_WATCH is not None else None

    def mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append(e)
        self.host.append(time.perf_counter())
        self.allocs.append(torch.cuda.memory_stats().get('num_device_alloc', 0))

    def stats(self):
        self.close()
        raw = [a.elapsed_time(b) for a, b in zip(self.events[:-1], self.events[1:])]
        if not raw:
            return None
        worst = max(range(len(raw)), key=lambda i: raw[i])
        host = [1e3 * (b - a) for a, b in zip(self.host[:-1], self.host[1:])]
        d = sorted(raw)
        n = len(d)
        med = d[n // 2] if n % 2 else 0.5 * (d[n // 2 - 1] + d[n // 2])
        return {'median': round(med, 3), 'min': round(d[0], 3), 'max': round(d[-1], 3), 'steps': n,
                'slowest_step': {'index': worst, 'device_ms': round(raw[worst], 3), 'host_ms_of_that_step': round(host[worst], 3),
                                 'host_ms_median': round(sorted(host)[len(host) // 2], 3)},
                'per_step_ms': [round(v, 2) for v in raw],
                'device_allocations_inside_the_loop': int(self.allocs1 - self.allocs0),
                'steps_with_a_device_allocation': [i for i, (a, b) in enumerate(zip(self.allocs[:-1], self.allocs[1:])) if b > a],
                'python_gc_inside_the_loop': self.gc(),
                'how': 'device-side time between the ends of consecutive steps (one event per step, read after the loop)'}


def allreduce_grads(params, world):
    """Data-parallel gradient exchange: ONE all-reduce over a flat fp32 bucket (RCCL ring over xGMI), after which
    every ``p.grad`` is a view of the averaged bucket.  Gradients are produced with ``p.grad = None`` before the
    backward pass, so autograd assigns them (no per-parameter accumulate kernels, no bucket memset)."""
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat)
    flat.div_(world)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


def make_reducer(params, world, args):
    """N > 1: persistent flat gradient buffer, buckets sent from autograd hooks while the backward pass still runs
    (sst_amd/parallel.py); --grad-sync flat = the round-2 behaviour (one blocking all-reduce after the backward pass)"""
    if world <= 1:
        return None
    from sst_amd.parallel import GradBucketReducer
    overlap = getattr(args, 'grad_sync', 'overlap') == 'overlap'
    return GradBucketReducer(params, n_buckets=getattr(args, 'grad_buckets', 2) if overlap else 1, overlap=overlap)


def collective_costs(reducer, dev, reps=20):
    """what the exchanges of one step cost when nothing overlaps them (HIP events around blocking calls, outside the timed
    region): `allreduce_ms` = the gradient buckets, `bn_sync_ms` = naiveSyncBN's statistics messages of one step
    (2 layers x (forward [mean || meansqr] of 2 C floats + backward 2 C floats), ops/norm.py:9-24, latency-bound)"""
    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    scratch = torch.zeros_like(reducer.flat)

    def grads():
        for start, end, _ in reducer.buckets:
            dist.all_reduce(scratch[start:end])

    small = [torch.zeros(2 * c, device=dev) for c in (64, 128)]

    def bn():
        for t in small:
            dist.all_reduce(t)     # forward statistics
        for t in small:
            dist.all_reduce(t)     # backward sums

    return {'allreduce_ms': round(timed(grads), 4), 'bn_sync_ms': round(timed(bn), 4),
            'gradient_bytes': int(reducer.flat.numel() * 4), 'buckets': [int((e - s_) * 4) for s_, e, _ in reducer.buckets]}


def remeasure_sra_traffic(args, timeout_s=240):
    """HBM traffic of the attention kernels measured NOW when rocprofv3 is on PATH (two PMC passes over tools/sra_only.py in a
    child process, after the timed region): -> path of the JSON, or None (no profiler, --no-traffic-remeasure, failure)."""
    import shutil
    import subprocess
    if args.no_traffic_remeasure or shutil.which('rocprofv3') is None or args.points != 116000 or args.frames_per_gpu != 1:
        return None
    out = os.path.join('gpurun_out', 'traffic_live')
    try:
        r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'collect_sra_traffic.sh'), out], cwd=ROOT, timeout=timeout_s,
                           capture_output=True, text=True, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
        path = os.path.join(ROOT, out, 'sra_traffic.json')
        return path if r.returncode == 0 and os.path.exists(path) else None
    except Exception:
        return None


def voxel_sort_key(coors):
    """(b, z, y, x) -> one int64 key whose order is the sorted-unique voxel order (bench-side helper: the reduced-precision
    leg compares two GPU runs and must not touch oracle/)"""
    c = coors.long()
    return ((c[:, 0] * 64 + c[:, 1]) * 4096 + c[:, 2]) * 4096 + c[:, 3]


def gpu_forward_sorted(model, frames, upstream_sorted=None):
    """Forward of the GPU pipeline without the voxel shuffle, rows re-ordered to the reference's sorted-unique voxel
    order: (features [M, C] on the host, int64 voxel keys [M] ascending).  With `upstream_sorted` ([M, C], rows in that
    sorted order) the pass also backpropagates it: parameter gradients are left in the model."""
    me = model.middle_encoder
    orig_shuffle = me.shuffle_voxels
    me.shuffle_voxels = False
    try:
        with torch.set_grad_enabled(upstream_sorted is not None):
            out = model(frames)
            key = voxel_sort_key(model.last_voxel_coors.cpu())
            order = torch.argsort(key)
            if upstream_sorted is not None:
                up = torch.empty_like(out)
                up[order.to(out.device)] = upstream_sorted.to(out.device)
                model.zero_grad(set_to_none=True)
                out.backward(up)
    finally:
        me.shuffle_voxels = orig_shuffle
    return out.detach().cpu()[order], key[order]


def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def cpu_reference_leg(model, frame_cpu, num_blocks, budget_s=75.0):
    """The CPU port of the reference data flow (oracle/cpu_pipeline.py; pinned against the reference ASSEMBLY by
    tests/test_ref_assembly.py, measured ratio to it in profiles/r03/cpu_ref_vs_port.json) beside the GPU path, on rank 0
    at N = 1:
      * thread sweep on BASELINE.json configs[0] (20 000 points, voxelize + DynamicScatter VFE + 1 SRA block, forward
        only; 1 warm-up + 3 timed, median) over {1, 8, 16, 32, 64, all}: every figure reported, the best named
        (round 2: 128 threads were SLOWER than 1 on this path);
      * thread choice for the headline frame: one forward pass of ONE block at full size per candidate {8, 16, 32, 64,
        all} (the op sizes of the real frame, a few seconds each);
      * `cpu_baseline`: frames/s of forward + backward on the bench frame itself at the chosen thread count - timed
        passes until `budget_s` of CPU time is spent (at most 3), median;
      * `parity`: the GPU forward AND backward of the SAME network (weights copied) on the SAME frame against the first
        timed CPU pass: kept-voxel sets equal, max abs feature error, relative error of parameter gradients at both ends of
        the network (north-star bar: 1e-3)."""
    from oracle.cpu_pipeline import CpuSSTBackbone, load_pipeline_weights   # the ONLY place bench.py touches oracle/
    all_threads = torch.get_num_threads()
    small = make_cloud(20000, 7, 'cpu')

    # BASELINE.json configs[0] with a thread sweep
    net1 = CpuSSTBackbone(VOXEL_SIZE, PC_RANGE, DROP_TRAIN, num_blocks=1).train()

    def timed_forward(net, cloud, nthreads, reps):
        torch.set_num_threads(nthreads)
        try:
            with torch.no_grad():
                net([cloud])                      # warm-up at this thread count
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    net([cloud])
                    ts.append(time.perf_counter() - t0)
        finally:
            torch.set_num_threads(all_threads)
        return _median(ts)

    sweep = sorted({t for t in (1, 8, 16, 32, 64, all_threads) if t <= all_threads})
    config0 = {str(t): round(1.0 / timed_forward(net1, small, t, 3), 4) for t in sweep}
    best0 = max(config0, key=lambda t: config0[t])
    # thread count for the headline frame: one block, forward, at full size
    probe = {str(t): round(timed_forward(net1, frame_cpu, t, 1), 3) for t in sweep if t >= 8 or t == all_threads}
    chosen = int(min(probe, key=lambda t: probe[t]))

    net = load_pipeline_weights(CpuSSTBackbone(VOXEL_SIZE, PC_RANGE, DROP_TRAIN, num_blocks=num_blocks).train(), model)
    torch.set_num_threads(chosen)
    try:
        times, out_first, up = [], None, None
        spent = 0.0
        while len(times) < 3 and (not times or spent + times[-1] < budget_s):
            net.zero_grad(set_to_none=True)
            t0 = time.perf_counter()
            out = net([frame_cpu])
            if up is None:
                up = torch.randn(out.shape, generator=torch.Generator().manual_seed(11))   # a fixed upstream gradient
            out.backward(up)
            times.append(time.perf_counter() - t0)
            spent += times[-1]
            if out_first is None:
                out_first = out.detach()
                grads_first = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        # the same port in float64 (integer stages in fp32): the adjudicator for `max_abs_err_vs_float64` - how far the GPU forward
        # is from the exact result of the network, in the headline arithmetic (dense products as the exact bf16 split) and with
        # the fp32 matrix pipe, beside the fp32 CPU port itself
        import copy
        net64 = copy.deepcopy(net).double()
        with torch.no_grad():
            out64 = net64([frame_cpu.double()])
        same64 = torch.equal(voxel_sort_key(net64.last_voxel_coors), voxel_sort_key(net.last_voxel_coors))
        del net64
    finally:
        torch.set_num_threads(all_threads)
    med = _median(times)

    # parity at the headline configuration: forward and backward
    dev = next(model.parameters()).device
    key_c = voxel_sort_key(net.last_voxel_coors)   # rows of the CPU output are in sorted-unique voxel order
    out_g, key_g = gpu_forward_sorted(model, [frame_cpu.to(dev)])
    voxels_equal = bool(key_c.numel() == key_g.numel() and torch.equal(key_c, key_g))
    vs64 = None
    if voxels_equal and same64:
        from sst_amd import dense as _dense
        mode_now = getattr(model.backbone, 'matmul', None) or _dense.matmul_mode()
        vs64 = {f'gpu, dense products {mode_now} (the timed mode)': float((out_g.double() - out64).abs().max()),
                'cpu port, fp32': float((out_first.double() - out64).abs().max())}
        other = 'f32' if mode_now != 'f32' else 'f32x6'
        names = {'f32': 'fp32', 'f32x6': 'f32x6', 'f32x3': 'f32x3'}      # the backbone keeps its own mode: switch it there
        try:
            model.backbone.set_precision(names[other])
            vs64[f'gpu, dense products {other}'] = float((gpu_forward_sorted(model, [frame_cpu.to(dev)])[0].double() - out64).abs().max())
        finally:
            model.backbone.set_precision(names[mode_now])
    grad_err = None
    if voxels_equal:
        gpu_forward_sorted(model, [frame_cpu.to(dev)], upstream_sorted=up)
        blocks = model.backbone.block_list
        pairs = {'vfe_layers.0.linear.weight': (model.voxel_encoder.vfe_layers[0].linear.weight, 'vfe.linears.0.weight'),
                 'block0.layer0.in_proj_weight': (blocks[0].encoder_list[0].win_attn.self_attn.in_proj_weight,
                                                  'layers.0.self_attn.in_proj_weight'),
                 f'block{num_blocks - 1}.layer1.linear2.weight': (blocks[-1].encoder_list[1].linear2.weight,
                                                                 f'layers.{2 * num_blocks - 1}.linear2.weight')}
        grad_err = {}
        for name, (p_gpu, cpu_name) in pairs.items():
            want = grads_first[cpu_name]
            grad_err[name] = float((p_gpu.grad.cpu() - want).abs().max() / want.abs().max().clamp(min=1e-12))
        model.zero_grad(set_to_none=True)
    parity = {'voxels_equal': voxels_equal, 'voxels': int(key_g.numel()),
              'max_abs_err': float((out_g - out_first).abs().max()) if voxels_equal else None,
              'max_abs_err_vs_float64': vs64,
              'max_rel_grad_err': grad_err, 'tolerance': 1e-3,
              'what': 'GPU forward + backward (fp32, no voxel shuffle, training-mode drop + batch-norm statistics, a fixed '
                      'random upstream gradient) vs the CPU port of the reference data flow with the same weights on the '
                      f'bench frame, all {num_blocks} SRA blocks; integer side = set of kept voxel coordinates; gradient '
                      'error = max |difference| / max |gradient| per parameter'}
    base = {'value': round(1.0 / med, 5), 'unit': 'frames/s', 'cores': chosen, 'kind': 'port',
            'sample': f'{len(times)} timed pass(es) of 1 frame ({frame_cpu.size(0)} points -> {out_first.size(0)} '
                      f'voxels), {num_blocks} SRA blocks, forward + backward, at {chosen} threads (fastest of a one-block '
                      f'forward probe at full size: {probe} s); median {med:.1f} s (passes: '
                      + ', '.join(f'{t:.1f}' for t in times) +
                      ' s); CPU port of the reference path (padded windows + nn.MultiheadAttention, '
                      'oracle/cpu_pipeline.py), pinned to the reference assembly by tests/test_ref_assembly.py',
            'host_threads': all_threads,
            'config0_20k_points_1_block_fwd': {'frames_per_s_by_threads': config0, 'best_threads': int(best0),
                                               'frames_per_s_best': config0[best0],
                                               'protocol': '1 warm-up + 3 timed per thread count, median'}}
    return base, parity


def config_as_is_leg(args, dev, frames, sync):
    """The detector a maintainer gets from INTEGRATION.md section A with the config UNTOUCHED: the
    resolved `model` of configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py (tests/golden/configs/, checked against the
    config text by tests/test_config_fixtures.py) through sst_amd.build_detector's class - no extra keyword (debug=True,
    reference_outputs default, no window_major, no mute), no set_precision(), no other call after construction - on the bench
    frame, forward + backward, each step calling the detector with the points only (no plan built ahead).  Reported:
      value / ms_per_step           the metric's scope: voxelize -> DynamicVFE -> SSTInputLayerV2 -> 6 SRA
                                    blocks, i.e. `det.extract_voxel_feats(points)` - comparable with the line's `value`;
      with_plan_prefetch            the same with the next batch's index plan queued behind the backward pass, as the main loop
                                    does (det.prepare / extract_voxel_feats(.., prepared=)): what a data-loader hook adds;
      full_backbone                 `det.extract_feat(points)` as the config says: + BEV canvas + the three attached
                                    convolutions with naiveSyncBN2d (MIOpen; solver search on during warm-up)."""
    import ast
    import contextlib
    path = os.path.join(ROOT, 'tests', 'golden', 'configs', 'sst_waymoD5_1x_3class_8heads_v2.model.py')
    cfg = ast.literal_eval(open(path).read())
    torch.manual_seed(0)
    det = Pipeline(model_cfg=cfg, voxel_feats_only=True).to(dev).train()
    params = [p for p in det.parameters() if p.requires_grad]
    seeds = {}

    def step(prepared=None):
        for p in params:
            p.grad = None
        out = det(frames, prepared)
        g = seeds.get(out.shape)
        if g is None:
            g = seeds[out.shape] = torch.randn(out.shape, device=out.device, dtype=out.dtype)
        out.backward(g)
        return out

    def timed(fn, warmup):
        for _ in range(warmup):
            fn()
        sync()
        st = StepTimes()
        st.mark()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
            st.mark()
        sync()
        el = time.perf_counter() - t0
        st.close()
        return {'value': round(args.frames_per_gpu * args.steps / el, 3), 'unit': 'frames/s',
                'ms_per_step': round(el / args.steps * 1e3, 3), 'steps': args.steps, 'step_ms': st.stats()}

    with contextlib.redirect_stdout(sys.stderr):      # mute=False as shipped: the layer prints its drop_info once
        plain = timed(step, max(3, args.warmup))
        ahead = []

        def pstep():
            out = step(ahead.pop() if ahead else None)
            ahead.append(det.prepare(frames, overlap=args.plan_overlap))
            return out
        pre = timed(pstep, 2)
        n_vox = int(det.last_voxel_coors.size(0))
        mode = {'matmul': det.backbone.matmul, 'precision': det.backbone.precision,
                'window_major': bool(det.middle_encoder.window_major), 'reference_outputs': bool(det.middle_encoder.reference_outputs),
                'debug': bool(det.middle_encoder.debug), 'fused_index_plan': det._planner is not None and det._planner is not False}
        full = None
        try:
            det.voxel_feats_only = False
            bench_flag = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            full = timed(step, 4)
        finally:
            det.voxel_feats_only = True
            torch.backends.cudnn.benchmark = bench_flag
    res = dict(plain)
    res.update(voxels_kept_per_gpu=n_vox, with_plan_prefetch=pre, full_backbone=full, defaults_in_effect=mode,
               config='configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py `model` (tests/golden/configs/'
                      'sst_waymoD5_1x_3class_8heads_v2.model.py), built with no extra keyword, no call after construction',
               what='sst_amd.DynamicVoxelNet.extract_voxel_feats(points) forward + backward per step (the metric\'s scope), the '
                    'points handed over each step with no plan built ahead')
    del det
    return res


def self_launch(n_ranks, line_out=None):
    """Re-run this script under torch.distributed.run with one process per GPU of this node."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: needed by RCCL on this driver stack
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_ranks}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks inherit the ORIGINAL stdout (this process's fd 1 is a copy of stderr by now: main())
    if line_out is not None:
        line_out.flush()
    return subprocess.call(cmd, env=env, stdout=line_out if line_out is not None else None)


def claim_stdout():
    """-> a file object on a private duplicate of the process's stdout; fd 1 itself and sys.stdout become copies of stderr, so
    that nothing but what is written to the returned object - the JSON line - reaches the caller's stdout, whoever prints
    (Python code, native libraries, child processes that inherit fd 1).  tests/test_bench_contract.py."""
    sys.stdout.flush()
    try:
        line_out = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    except OSError:              # no usable descriptors (embedded interpreter): the Python-level redirection alone
        line_out = sys.stdout
    sys.stdout = sys.stderr
    return line_out


def main():
    # the contract: rank 0 prints ONE JSON line.  Modules built from a shipped config print (the input layer announces its
    # drop_info unless `mute` is set): everything but the line goes to stderr.
    # ... at the level of the file descriptor too: native libraries write to fd 1 themselves (gloo announces its peers there when a
    # process group is created; RCCL's diagnostics go there when NCCL_DEBUG is set) - fd 1 becomes a copy of stderr, the line is
    # written to a private duplicate of the original stdout.
    args = parse_args()          # --help goes to the real stdout
    python_stdout = sys.stdout
    line_out = claim_stdout()
    try:
        _main(args, line_out)
    finally:
        line_out.flush()
        sys.stdout = python_stdout


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--frames-per-gpu', type=int, default=1)
    ap.add_argument('--points', type=int, default=None, help='points per frame (default: 116000 for the SST workloads)')
    ap.add_argument('--workload', default='sst', choices=('sst', 'sst_bs2', 'sst_bev', 'sst_center', 'fsd', 'fsdv2'),
                    help="sst_center = configs/sst_refactor/sst_waymoD5_1x_3class_centerhead.py as shipped (4 SRA blocks with COSINE "
                         "attention, checkpoint_blocks=[0, 1], 5-channel points; voxel features scope; beside it the same config "
                         "with standard attention); "
                         "sst = the headline (BASELINE.json configs[1]/[2] geometry, 1 frame/GPU); sst_bs2 = configs[2]'s "
                         "2 frames per GPU; sst_bev = sst + recover_bev + the three attached convolutions of the config; "
                         "fsd / fsdv2 = configs[3] / configs[4] hot paths (bench_workloads.py)")
    ap.add_argument('--blocks', type=int, default=6)
    ap.add_argument('--fwd-only', action='store_true', help='also report nothing else; time the forward only')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-plan-prefetch', action='store_true',
                    help='build the index plan of a batch at the head of its own step (round-2 behaviour) instead of behind '
                         'the previous step\'s backward pass')
    ap.add_argument('--plan-overlap', action='store_true',
                    help='build the next batch\'s index plan on the planner\'s own stream beside the backward pass '
                         '(FramePlanner.build_overlapped) instead of on the same stream behind it.  Measured without a profiler '
                         '(tools/step_segments.py, profiles/r05): 11.50 -> 11.47 ms on the bench frame, 5.07 -> 4.97 ms on the '
                         'LiDAR-like frame - the head of a step is NOT where the time goes; off by default')
    ap.add_argument('--grad-sync', default='overlap', choices=('overlap', 'flat'),
                    help='N > 1: overlap = buckets sent from autograd hooks during the backward pass (default); flat = one '
                         'blocking all-reduce of the whole buffer after it')
    ap.add_argument('--grad-buckets', type=int, default=2)
    ap.add_argument('--no-time-sra-bwd', action='store_true', help='do not attach events to the SRA backward launches')
    ap.add_argument('--no-forward-only-leg', action='store_true', help='skip the extra forward-only measurement')
    ap.add_argument('--no-f32x3-leg', action='store_true', help='skip the split-precision (bf16 x 3) leg beside the headline')
    ap.add_argument('--conv-precision', choices=['f32', 'f32x6', 'f32x3'], default='f32x6',
                    help="--workload fsd | fsdv2: how the sparse convolutions (forward + data gradient) multiply in the TIMED steps: "
                         "'f32x6' (default) = exact three-way bf16 split, six products (csrc/spconv_os_x6.hip; admissible as exact fp32: "
                         "tests/test_gpu_spconv.py), 'f32' = the fp32 matrix pipe, 'f32x3' = the two-way split (a leg, never the "
                         "headline: the line then says so in dtype)")
    ap.add_argument('--no-traffic-remeasure', action='store_true',
                    help='take roofline.traffic from profiles/latest_sra_traffic.json instead of two rocprofv3 --pmc passes now')
    ap.add_argument('--cloud', default='uniform', choices=('uniform', 'lidar'),
                    help='diagnostic: lidar = the MAIN loop on the LiDAR-like frame of the lidar_like_cloud leg (for profiles of '
                         'that frame; the line then names it in config.workload and is not the headline)')
    ap.add_argument('--no-lidar-leg', action='store_true', help='skip the LiDAR-like / pathological frame beside the headline')
    ap.add_argument('--no-bf16-leg', action='store_true', help='skip the reduced-precision (bf16) measurement')
    ap.add_argument('--no-std-attention-leg', action='store_true',
                    help='--workload sst_center: skip the same config with standard attention beside it')
    ap.add_argument('--no-bf16-own-process', action='store_true',
                    help='do not repeat the reduced-precision leg as the main loop of a process of its own')
    ap.add_argument('--matmul', default='f32x6', choices=('f32', 'f32x6'),
                    help="how the fp32 encoder layers multiply in the TIMED region: 'f32x6' = exact three-way bf16 split, six "
                         "products on the bf16 matrix pipe (csrc/dense_f32x6.hip; admissible as exact fp32: tests/"
                         "test_gpu_dense_f32x6.py) - the library's default, nothing is switched; 'f32' = the fp32 matrix pipe "
                         "(csrc/dense_f32.hip), via set_precision('fp32')")
    ap.add_argument('--no-config-as-is-leg', action='store_true',
                    help='skip the leg that builds the detector from the shipped config (tests/golden/configs/) with no extra '
                         'keyword and no call after construction')
    ap.add_argument('--precision', default='f32', choices=('f32', 'bf16'),
                    help="precision of the encoder layers in the TIMED region; the contract's headline is f32 (default), "
                         "bf16 makes the reduced-precision mode the measured one (profiling)")
    ap.add_argument('--piecewise-index', action='store_true',
                    help='index plan through the module interfaces (three read-backs) instead of csrc/frame_plan.hip')
    ap.add_argument('--impl', type=int, default=0, help='0 = MFMA SRA kernels, 1 = generic VALU kernels')
    ap.add_argument('--backend', default='nccl', help='nccl (= RCCL, default) | gloo (dev check of the N>1 path on one GPU)')
    ap.add_argument('--share-device', action='store_true', help='dev only: every rank uses cuda:0')
    ap.add_argument('--compact', action='store_true',
                    help='a short leg of another workload inside the default line: no side legs, no CPU baseline; the '
                         'FSD / FSDv2 chains check parity against the CPU port on a bounded frame (--parity-points)')
    ap.add_argument('--parity-points', type=int, default=40000,
                    help='--compact, fsd / fsdv2: points of the frame the GPU chain is compared with the CPU port on')
    ap.add_argument('--no-workload-legs', action='store_true',
                    help='default line only: skip the compact legs of BASELINE configs 2-4 (sst_bs2, sst_center, fsd, fsdv2)')
    ap.add_argument('--no-voxelize-roofline', action='store_true', help='skip the dynamic-voxelize microbenchmark')
    ap.add_argument('--no-gemm-tuning', action='store_true',
                    help='do not let PyTorch TunableOp pick the hipBLASLt/rocBLAS solution of each dense GEMM shape')
    return ap.parse_args()


def voxelize_roofline(dev, frame_points):
    """dynamic voxelize alone (24 B / point = 12 B xyz in + 12 B zyx out) at 16 M points - the size
    at which the launch is bound by HBM - and at the bench frame's size, where one launch is a few microseconds of latency."""
    import sst_amd
    vs, rng = (0.32, 0.32, 6), [-74.88, -74.88, -2, 74.88, 74.88, 4]
    out = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'algorithmic_bytes_per_point': 24,
           'kernel': 'dynamic_voxelize_rows_k<3> (csrc/voxelize.hip), one launch per sst_dynamic_voxelize_f32 call; '
                     'torch events on the launch stream, 20 launches'}
    for key, n in (('at_16M_points', 1 << 24), ('at_bench_frame', int(frame_points))):
        g = torch.Generator(device=dev).manual_seed(5)
        pts = torch.rand(n, 3, device=dev, generator=g) * torch.tensor([149.76, 149.76, 6.0], device=dev) \
            + torch.tensor([-74.88, -74.88, -2.0], device=dev)
        for _ in range(3):
            sst_amd.voxelization(pts, vs, rng, -1, -1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            sst_amd.voxelization(pts, vs, rng, -1, -1)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        out[key] = {'points': n, 'avg_us': round(us, 2), 'achieved': round(24.0 * n / us / 1e3, 1),
                    'frac': round(24.0 * n / us / 1e3 / HBM_PEAK_GBS, 4)}
        del pts
    out['achieved'], out['frac'] = out['at_16M_points']['achieved'], out['at_16M_points']['frac']
    out['note'] = ('frac is quoted at 16 M points; at the bench frame one launch moves 2.8 MB and is bound by launch latency, '
                   'not by bandwidth (at_bench_frame)')
    return out


def workload_legs(args):
    """BASELINE.json configs 2-4 as compact legs of the default line, each the MAIN loop of a process of its own (fresh
    interpreter, allocator, clocks; this process idle meanwhile): value, step statistics, host time, the leg's roofline, and for
    the FSD chains parity against the CPU port on a bounded frame.  VERDICT round 5 item 2: every BASELINE config is then in the
    line the driver records, not only in builder-run files."""
    import subprocess
    legs = {}
    # fsd_bs2: the FSD chain at 2 frames per GPU, the batch its config trains at (configs/fsd/fsd_waymoD1_1x.py: samples_per_gpu=2):
    # the segmented reduce of the SIR layers, launch-bound at one frame (18 k foreground points), is quoted there too; no CPU pass
    plan = (('sst_bs2', 10, 3, 150, ()), ('sst_center', 10, 3, 150, ()), ('fsd', 10, 3, 240, ()),
            ('fsd_bs2', 10, 3, 240, ('--workload', 'fsd', '--frames-per-gpu', '2', '--no-cpu-baseline')), ('fsdv2', 10, 3, 300, ()))
    for name, steps, warm, limit, extra in plan:
        note(f'leg: workload {name} (own process)')
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', name, '--compact', '--steps', str(steps), '--warmup',
               str(warm)]
        if extra:
            cmd = [sys.executable, os.path.abspath(__file__), *extra, '--compact', '--steps', str(steps), '--warmup', str(warm)]
        t0 = time.perf_counter()
        try:
            sub = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, cwd=ROOT)
            own = json.loads(sub.stdout.strip().splitlines()[-1])
            leg = {k: own.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'step_ms', 'host_ms_per_step',
                                           'dtype') if k in own}
            leg['workload'] = (own.get('config') or {}).get('workload')
            for k in ('frames_per_gpu', 'points_per_frame', 'voxels_per_gpu', 'sizes'):
                if k in (own.get('config') or {}):
                    leg[k] = own['config'][k]
            rf = own.get('roofline')
            if rf:
                leg['roofline'] = {k: rf.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_us',
                                                          'peak_note', 'frac_of_bf16_peak') if k in rf}
                if isinstance(rf.get('sra_bwd'), dict):
                    leg['roofline']['sra_bwd'] = {k: rf['sra_bwd'].get(k) for k in ('achieved', 'frac', 'avg_us') if k in rf['sra_bwd']}
            if own.get('roofline_seg_reduce'):
                leg['roofline_seg_reduce'] = {k: own['roofline_seg_reduce'].get(k) for k in ('bound', 'achieved', 'peak', 'unit',
                                                                                             'frac', 'algorithmic_bytes', 'ms')}
            if own.get('reduced_precision'):
                rp = own['reduced_precision']
                leg['reduced_precision'] = {k: rp.get(k) for k in ('value', 'ms_per_step', 'roofline') if k in rp}
            par = own.get('parity')
            if par:
                leg['parity'] = {k: par.get(k) for k in ('integer_outputs_equal', 'max_abs_err_overall', 'max_abs_err',
                                                         'max_rel_grad_err', 'feature_tolerance', 'frame_points', 'unpinned')
                                 if k in par}
            leg['wall_s'] = round(time.perf_counter() - t0, 1)
            legs[name] = leg
        except Exception as e:     # a leg beside the headline must never take the line down
            legs[name] = {'error': repr(e)[:300], 'wall_s': round(time.perf_counter() - t0, 1)}
    return legs


def _main(args, line_out):
    global GC_WATCH
    GC_WATCH = GcWatch()
    args.points_given = args.points is not None
    if args.points is None:
        args.points = 116000
    if args.workload == 'sst_bs2':
        args.frames_per_gpu = 2
    if args.compact:
        args.no_lidar_leg = args.no_forward_only_leg = args.no_config_as_is_leg = args.no_traffic_remeasure = True
        args.no_f32x3_leg = args.no_bf16_own_process = args.no_std_attention_leg = args.no_workload_legs = True
        args.no_voxelize_roofline = True
        if args.workload not in ('fsd', 'fsdv2'):
            args.no_cpu_baseline = True

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # launched as plain `python bench.py --gpus N`: start one rank per GPU ourselves, the way the reference's
        # tools/dist_train.sh:7-9 does (torch.distributed.launch --nproc_per_node=$GPUS); rank 0 prints the JSON line
        sys.exit(self_launch(args.gpus, line_out))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    if args.workload in ('fsd', 'fsdv2'):
        import bench_workloads
        bench_workloads.run(args, rank, world, dev, make_reducer, line_out=line_out, step_times_cls=StepTimes)
        if world > 1:
            dist.destroy_process_group()
        return

    from sst_amd import kernels as K
    # (no TunableOp here since round 6: the SST step has had no library GEMM in it since round 4 - every dense product is a
    # kernel of csrc/ - so there was nothing for it to tune; the FSD workloads, whose stand-in heads are library GEMMs, enable
    # it in bench_workloads.run)
    torch.manual_seed(0)                      # identical initial weights on every rank
    point_channels = 3
    if args.workload == 'sst_center':
        center_cfg = load_config_fixture('sst_waymoD5_1x_3class_centerhead')
        args.blocks = int(center_cfg['backbone']['num_blocks'])
        point_channels = int(center_cfg['voxel_encoder']['in_channels'])
        model = Pipeline(model_cfg=center_cfg, voxel_feats_only=True).to(dev)
        args.no_cpu_baseline = args.no_lidar_leg = args.no_f32x3_leg = args.no_config_as_is_leg = True
        args.no_bf16_own_process = True      # the reduced-precision leg stays: cosine attention inside the bf16 kernels too
    else:
        model = Pipeline(args.blocks, with_bev=args.workload == 'sst_bev').to(dev)
    if args.workload == 'sst_bev':
        torch.backends.cudnn.benchmark = True    # MIOpen: search for the convolution solvers during warm-up
        args.no_bf16_leg = args.no_cpu_baseline = args.no_lidar_leg = args.no_f32x3_leg = True   # those legs cover the voxel features only
    model.train()
    model.backbone.set_impl(args.impl)
    model.fused_index = not args.piecewise_index
    if args.precision == 'bf16':
        model.backbone.set_precision('bf16')
        args.no_bf16_leg = True
    elif args.matmul == 'f32':
        model.backbone.set_precision('fp32')     # the opt-out: 'f32x6' (exact split) is the library's default, no call needed
    params = [p for p in model.parameters() if p.requires_grad]
    frames = [make_cloud(args.points, 1000 * rank + i, dev, point_channels) for i in range(args.frames_per_gpu)]
    if args.cloud == 'lidar':
        frames = [make_lidar_cloud(2000 * rank + i, dev) for i in range(args.frames_per_gpu)]
        args.points = int(frames[0].size(0))
        args.no_lidar_leg = args.no_cpu_baseline = True
    torch.manual_seed(1234 + rank)            # per-rank voxel shuffles
    reducer = make_reducer(params, world, args)

    seed_grad = {}
    ahead = []      # the index plan of the NEXT batch (depends on the point clouds only: no parameters, no features)

    def next_plan():
        """Software pipelining across steps, as a data loader's prefetch would do it: the index plan of the next batch
        (voxelize, sorted-unique, window bucketing: ~50 short launches) is queued on the SAME stream right behind this
        step's backward pass, while the host is still ahead of the device - its launches then run back to back instead of
        at the host's launch rate at the head of the next step.  Every timed step still builds exactly one plan."""
        if args.no_plan_prefetch:
            return None
        return ahead.pop() if ahead else model.prepare(frames, overlap=args.plan_overlap)

    def host_ms_per_step(n=3):
        """Interpreter + launch time of one step: forward and backward each timed on the host with the device idle at their
        start (a full synchronisation in between), so that nothing waits for a kernel except the forward's one size
        read-back.  A step whose kernels take less than this is bound by the host, whatever the kernels do."""
        if args.fwd_only:
            return None
        tot = 0.0
        for _ in range(n):
            for p in params:
                p.grad = None
            sync()
            t_a = time.perf_counter()
            out = model(frames)
            t_b = time.perf_counter()
            g = seed_grad.get(out.shape)
            if g is None:
                g = seed_grad[out.shape] = torch.randn(out.shape, device=out.device, dtype=out.dtype)
            sync()
            t_c = time.perf_counter()
            out.backward(g)
            t_d = time.perf_counter()
            sync()
            if reducer is not None:
                reducer.finish()
            tot += (t_b - t_a) + (t_d - t_c)
        return round(tot / n * 1e3, 3)

    def fresh_allocator():
        """Every leg beside the headline starts from an empty caching allocator: the legs keep differently sized tensors (one
        512 MB slab per layer in the exact-split mode, ~25 tensors per layer in the others, half-size ones in the bf16 mode), and
        a leg that inherited the previous one's cached blocks spent its timed steps splitting and re-requesting segments
        (bf16 leg 94-112 frames/s behind another leg, 163 as the main loop of its own process)."""
        sync()
        if not os.environ.get('SST_BENCH_NO_EMPTY_CACHE'):
            torch.cuda.empty_cache()
        # ... and from a collector that has nothing old to walk: the objects the previous legs left (a whole second model after
        # the config_as_is leg) join the permanent generation, as after the main loop's warm-up.  Without it the reduced-precision
        # leg carried ONE step of 38 ms among twenty of 6.1 (a full collection inside its timed loop: 129 frames/s in the line,
        # 160 as the main loop of a process of its own - profiles/r05; SST_BENCH_NO_LEG_GC_FREEZE=1 reproduces it).
        if not os.environ.get('SST_BENCH_NO_LEG_GC_FREEZE'):
            import gc as _gc
            _gc.collect()
            _gc.freeze()

    def step():
        if args.fwd_only:
            with torch.no_grad():
                out = model(frames, next_plan())
                if not args.no_plan_prefetch:
                    ahead.append(model.prepare(frames, overlap=args.plan_overlap))
                return out
        for p in params:
            p.grad = None
        out = model(frames, next_plan())
        # backward from a fixed random upstream gradient (what a detection head would hand back).  NOT out.sum(): the
        # sum over the channels of a LayerNorm output is a constant, so its gradient is exactly zero upstream of the
        # last LayerNorm and every backward kernel would be timed on all-zero operands (lower power, higher clocks)
        g = seed_grad.get(out.shape)
        if g is None:
            g = seed_grad[out.shape] = torch.randn(out.shape, device=out.device, dtype=out.dtype)
        out.backward(g)                       # the reducer's hooks send a bucket as soon as its gradients exist
        if not args.no_plan_prefetch:
            ahead.append(model.prepare(frames, overlap=args.plan_overlap))
        if reducer is not None:
            reducer.finish()
        return out

    note(f'warm-up: {args.warmup} steps')
    # What a training script does once its model and data pipeline are built: collect, then move everything alive to the permanent
    # generation.  A full collection of a process that has imported torch walks ~10^6 objects (60-70 ms, measured: one step of
    # 71 ms among twenty of 6.5 ms in the reduced-precision leg; none with the collector off); frozen objects are not walked, the
    # collector stays ON for what the steps allocate.  It happens INSIDE the warm-up (before its last two steps), not between the
    # warm-up and the timed loop: the device idles for those 60-70 ms and drops its clocks, and the first timed step then ran
    # 2.5 ms longer than the other nineteen (step_ms.slowest_step.index = 0 in every round-5 / early round-6 line).
    import gc
    tail_steps = min(2, max(args.warmup - 1, 0))
    for _ in range(args.warmup - tail_steps):
        out = step()
    gc.collect()
    gc.freeze()
    for _ in range(tail_steps):
        out = step()
    note('timed loop')
    n_voxels = int(model.last_voxel_coors.size(0))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sink = []
    K.EVENT_SINK = sink
    # HIP events bound to every 5th SRA forward launch and every 5th one-pass SRA backward launch of the timed region
    # (12 launches of each per step, both window shifts get sampled)
    K.EVENT_STRIDE = 5
    K.EVENT_KINDS = ('sra_fwd',) if (args.fwd_only or args.no_time_sra_bwd) else ('sra_fwd', 'sra_bwd')
    if args.precision == 'bf16':
        K.EVENT_KINDS = ()
    if os.environ.get('SST_BENCH_ALLOC_TRACE_TIMED'):     # diagnostic: who calls hipMalloc inside the timed loop
        torch.cuda.memory._record_memory_history(enabled='all', context=None, stacks='python', max_entries=200000)
    sync()
    main_times = StepTimes()
    main_times.mark()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        main_times.mark()
    sync()
    elapsed = time.perf_counter() - t0
    main_times.close()
    if os.environ.get('SST_BENCH_ALLOC_TRACE_TIMED'):
        snap = torch.cuda.memory._snapshot()
        for e in [e for tr in snap.get('device_traces', []) for e in tr if e.get('action') in ('segment_alloc', 'segment_free')][:20]:
            fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in (e.get('frames') or [])[:12]]
            print('timed-loop', e['action'], e['size'], 'B', fr, file=sys.stderr)
        torch.cuda.memory._record_memory_history(enabled=None)
    K.EVENT_SINK = None
    per_rank = None
    if world > 1:
        # every rank's own clock around the same K steps (the line's time is their maximum), so that a scaling run describes
        # itself: a straggler rank shows up here, not as an unexplained loss of efficiency
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ms = [float(v.item()) / args.steps * 1e3 for v in every]
        per_rank = {'ms_per_step': [round(v, 3) for v in ms], 'min': round(min(ms), 3), 'max': round(max(ms), 3)}
        elapsed = max(float(v.item()) for v in every)

    if os.environ.get('SST_BENCH_ALLOC_TRACE'):
        # diagnostic: which device allocations (hipMalloc through torch's caching allocator) happen inside a steady-state step
        keys = ('num_device_alloc', 'num_device_free', 'reserved_bytes.all.current', 'active_bytes.all.current', 'num_alloc_retries')
        try:
            torch.cuda.memory._record_memory_history(enabled='all', context=None, stacks='python', max_entries=200000)
        except Exception as e:
            print('memory history unavailable:', e, file=sys.stderr)
        for i in range(3):
            before = {k: torch.cuda.memory_stats().get(k, 0) for k in keys}
            step()
            sync()
            after = {k: torch.cuda.memory_stats().get(k, 0) for k in keys}
            print('alloc trace step', i, {k: after[k] - before[k] for k in keys}, 'reserved MB', after['reserved_bytes.all.current'] >> 20,
                  file=sys.stderr)
        try:
            snap = torch.cuda.memory._snapshot()
            ev = [e for tr in snap.get('device_traces', []) for e in tr if e.get('action') in ('segment_alloc', 'segment_free')]
            for e in ev[:60]:
                fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in (e.get('frames') or [])[:6]]
                print('  ', e['action'], e['size'] >> 20, 'MB', fr, file=sys.stderr)
            torch.cuda.memory._record_memory_history(enabled=None)
        except Exception as e:
            print('snapshot failed:', e, file=sys.stderr)

    comm = collective_costs(reducer, dev) if reducer is not None else None   # every rank takes part
    main_host_ms = host_ms_per_step() if world == 1 else None                 # outside the timed region

    # Outside the timed region: the forward-only rate of the same workload (BASELINE.json configs[1] is quoted
    # forward-only, the metric forward + backward; `value` is the harder one, this is reported beside it).
    fwd_only = None
    note('main loop done; side legs')
    if not args.fwd_only and not args.no_forward_only_leg:
        note('leg: forward only')
        with torch.no_grad():
            for _ in range(2):
                model(frames)
            sync()
            fo_times = StepTimes()
            fo_times.mark()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                model(frames)
                fo_times.mark()
            sync()
            el = time.perf_counter() - t1
            fo_times.close()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        fwd_only = {'value': round(world * args.frames_per_gpu * args.steps / el, 3), 'unit': 'frames/s',
                    'ms_per_step': round(el / args.steps * 1e3, 3), 'steps': args.steps, 'step_ms': fo_times.stats(),
                    'note': 'same pipeline under torch.no_grad(), training-mode drop/shuffle; not part of `value`'}

    # --workload sst_center: the same config with standard attention (layer_cfg cosine off), same frames, same loop: what the
    # normalisation + per-head temperature inside the kernels cost per encoder layer
    std_leg = None
    if args.workload == 'sst_center' and not args.fwd_only and world == 1 and not args.no_std_attention_leg:
        import copy
        fresh_allocator()
        std_cfg = copy.deepcopy(center_cfg)
        std_cfg['backbone']['layer_cfg'] = dict(std_cfg['backbone']['layer_cfg'], cosine=False)
        torch.manual_seed(0)
        keep_model, keep_params = model, params
        model = Pipeline(model_cfg=std_cfg, voxel_feats_only=True).to(dev).train()
        params = [p for p in model.parameters() if p.requires_grad]
        ahead.clear()
        try:
            for _ in range(max(3, args.warmup)):
                step()
            sync()
            sd_times = StepTimes()
            sd_times.mark()
            t5 = time.perf_counter()
            for _ in range(args.steps):
                step()
                sd_times.mark()
            sync()
            el = time.perf_counter() - t5
            sd_times.close()
        finally:
            model, params = keep_model, keep_params
            ahead.clear()
        n_layers = 2 * args.blocks
        std_ms = el / args.steps * 1e3
        cos_ms = elapsed / args.steps * 1e3
        std_leg = {'value': round(world * args.frames_per_gpu * args.steps / el, 3), 'unit': 'frames/s',
                   'ms_per_step': round(std_ms, 3), 'steps': args.steps, 'step_ms': sd_times.stats(),
                   'cosine_minus_std_ms_per_layer': round((cos_ms - std_ms) / n_layers, 4),
                   'cosine_over_std': round(cos_ms / std_ms, 4),
                   'what': 'the same shipped config with layer_cfg.cosine = False (standard scaled-dot-product attention), same '
                           f'frames and loop; {n_layers} encoder layers, blocks 0-1 recomputed in the backward pass (checkpoint_blocks)'}

    as_is = None
    if (world == 1 and not args.fwd_only and not args.no_config_as_is_leg and args.workload == 'sst' and args.cloud == 'uniform'
            and args.blocks == 6 and args.precision == 'f32'):
        fresh_allocator()
        note('leg: config_as_is')
        as_is = config_as_is_leg(args, dev, frames, sync)
        as_is['vs_value'] = None     # filled in below

    # Order of the legs: the reduced-precision leg right behind the headline and the forward-only rate, the fp32-matrix-pipe leg
    # last - it draws the most power of all, and on some boxes whatever ran behind it ran at lower clocks (bf16 leg 112 frames/s
    # behind it, 163 as the main loop of the same process minutes later).
    # Beside the fp32 headline: the same step with the encoder layers in the reduced-precision mode (bf16 storage, fp32
    # accumulation / softmax / LayerNorm statistics, fp32 master weights: sst_amd/bf16.py) - what the reference's own
    # fp16 training of these layers (configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82) corresponds to here.
    bf16_leg = None
    if not args.fwd_only and not args.no_bf16_leg:
        note('leg: reduced precision (bf16)')
        fresh_allocator()
        with torch.no_grad():
            ref_out, ref_key = gpu_forward_sorted(model, frames)
        model.backbone.set_precision('bf16')
        try:
            with torch.no_grad():
                low_out, low_key = gpu_forward_sorted(model, frames)
            diff = (low_out - ref_out).abs() if torch.equal(ref_key, low_key) else None
            for _ in range(max(3, args.warmup)):
                step()
            bsink = []
            K.EVENT_SINK = bsink
            K.EVENT_KINDS = () if os.environ.get('SST_BENCH_BF16_NO_EVENTS') else ('sra_fwd_bf16', 'sra_bwd_bf16')
            sync()
            t2 = time.perf_counter()
            per_step = []
            dev_allocs0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
            bf_times = StepTimes()
            bf_times.mark()
            for _ in range(args.steps):
                ts = time.perf_counter()
                step()
                bf_times.mark()
                per_step.append(time.perf_counter() - ts)
            sync()
            el = time.perf_counter() - t2
            bf_times.close()
            if os.environ.get('SST_BENCH_DEBUG'):
                print('bf16 leg host time per step (ms):', [round(1e3 * v, 2) for v in per_step], 'device allocations during the leg:',
                      torch.cuda.memory_stats().get('num_device_alloc', 0) - dev_allocs0, file=sys.stderr)
            K.EVENT_SINK = None
            bf16_host_ms = host_ms_per_step()      # still in the bf16 mode
        finally:
            model.backbone.set_precision('f32x6' if args.matmul == 'f32x6' else 'fp32')
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())

        def bstats(kind, bytes_per_token):
            ev = [(e0.elapsed_time(e1), n) for k_, e0, e1, n in bsink if k_ == kind]
            ev = [(t_, n) for t_, n in ev if t_ > 0]
            if not ev:
                return None
            ms_ = sum(t_ for t_, _ in ev) / len(ev)
            tok_ = sum(n for _, n in ev) / len(ev)
            ach = bytes_per_token * tok_ / (ms_ * 1e-3) / 1e9
            return {'achieved': round(ach, 1), 'frac': round(ach / HBM_PEAK_GBS, 4), 'avg_launch_ms': round(ms_, 4),
                    'algorithmic_bytes_per_launch': int(bytes_per_token * tok_), 'launches_timed': len(ev)}

        bf16_leg = {'value': round(world * args.frames_per_gpu * args.steps / el, 3), 'unit': 'frames/s',
                    'ms_per_step': round(el / args.steps * 1e3, 3), 'steps': args.steps, 'dtype': 'bf16',
                    'step_ms': bf_times.stats(),
                    'what': 'same step, encoder layers in bf16 storage (fp32 accumulate / softmax / LayerNorm statistics, '
                            'fp32 master weights); voxelize, VFE and the index plan unchanged (fp32, as the reference '
                            'forces them: voxel_encoder.py:229)',
                    'roofline': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'sra_fwd': bstats('sra_fwd_bf16', 4 * 128 * 2 + 8),
                                 'sra_bwd': bstats('sra_bwd_bf16', 8 * 128 * 2 + 8)},
                    'vs_fp32_forward': None if diff is None else {'max_abs': float(diff.max()), 'mean_abs': float(diff.mean()),
                                                                  'voxels_equal': True}}
        bf16_leg['host_ms_per_step'] = bf16_host_ms
        bf16_leg['host_bound'] = bool(bf16_host_ms is not None and bf16_host_ms > 0.9 * bf16_leg['ms_per_step'])
        # The same leg as the MAIN loop of a process of its own (fresh interpreter, allocator and clocks; this process idle
        # meanwhile): if the two disagree, state inherited from the legs in front of it is the cause, if they agree and the
        # driver's box still differs from the builder's, it is the box.  Both are in the line.
        if world == 1 and not args.no_bf16_own_process:
            note('leg: reduced precision, own process')
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--precision', 'bf16', '--steps', str(args.steps), '--warmup',
                   str(max(3, args.warmup)), '--points', str(args.points), '--blocks', str(args.blocks), '--no-cpu-baseline',
                   '--no-lidar-leg', '--no-forward-only-leg', '--no-config-as-is-leg', '--no-traffic-remeasure', '--no-f32x3-leg',
                   '--no-time-sra-bwd', '--no-voxelize-roofline', '--no-workload-legs']
            try:
                sub = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
                own = json.loads(sub.stdout.strip().splitlines()[-1])
                bf16_leg['own_process'] = {k: own.get(k) for k in ('value', 'ms_per_step', 'step_ms', 'host_ms_per_step')}
                bf16_leg['own_process']['vs_in_process'] = round(own['value'] / bf16_leg['value'], 4)
            except Exception as e:     # the leg beside the headline must never take the line down
                bf16_leg['own_process'] = {'error': repr(e)[:200]}

    # Beside the headline (uniform cloud): the same step on a LiDAR-like frame with out-of-range points and duplicates
    lidar_leg = None
    if not args.fwd_only and not args.no_lidar_leg:
        note('leg: LiDAR-like cloud')

        def lidar_run(nframes):
            """W warm-up + K timed steps on `nframes` LiDAR-like frames per GPU; the attention kernels' own launch times (events
            bound to every 5th launch) against their algorithmic bytes, as for the headline"""
            fresh_allocator()
            lframes = [make_lidar_cloud(2000 * rank + i, dev) for i in range(nframes)]

            lahead = []     # the next step's index plan queued behind this step's backward pass, exactly as the main loop does it

            def lstep():
                for p in params:
                    p.grad = None
                o = model(lframes, None if args.no_plan_prefetch else (lahead.pop() if lahead else model.prepare(lframes)))
                gg = seed_grad.get(o.shape)
                if gg is None:
                    gg = seed_grad[o.shape] = torch.randn(o.shape, device=o.device, dtype=o.dtype)
                o.backward(gg)
                if not args.no_plan_prefetch:
                    lahead.append(model.prepare(lframes))
                if reducer is not None:
                    reducer.finish()
                return o

            for _ in range(3):
                lo = lstep()
            lsink = []
            K.EVENT_SINK = lsink
            K.EVENT_STRIDE = 5
            K.EVENT_KINDS = ('sra_fwd', 'sra_bwd')
            sync()
            li_times = StepTimes()
            li_times.mark()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                lstep()
                li_times.mark()
            sync()
            el = time.perf_counter() - t3
            li_times.close()
            K.EVENT_SINK = None
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            sizes = []
            for sft in range(2):
                pl = model.last_plans[sft]
                if pl is not None:
                    off = pl.winoff[:pl.n_windows + 1].cpu()
                    d = (off[1:] - off[:-1]).float()
                    sizes.append({'windows': int(pl.n_windows), 'tokens_min': int(d.min()), 'tokens_mean': round(float(d.mean()), 1),
                                  'tokens_max': int(d.max())})

            def lstats(kind, bytes_per_token):
                ev = [(e0.elapsed_time(e1), n) for k_, e0, e1, n in lsink if k_ == kind]
                ev = [(t_, n) for t_, n in ev if t_ > 0]
                if not ev:
                    return None
                ms_ = sum(t_ for t_, _ in ev) / len(ev)
                tok_ = sum(n for _, n in ev) / len(ev)
                ach = bytes_per_token * tok_ / (ms_ * 1e-3) / 1e9
                return {'achieved': round(ach, 1), 'frac': round(ach / HBM_PEAK_GBS, 4), 'avg_launch_ms': round(ms_, 4),
                        'algorithmic_bytes_per_launch': int(bytes_per_token * tok_), 'launches_timed': len(ev)}

            return {'value': round(world * nframes * args.steps / el, 3), 'unit': 'frames/s', 'frames_per_gpu': nframes,
                    'ms_per_step': round(el / args.steps * 1e3, 3), 'ms_per_frame': round(el / args.steps / nframes * 1e3, 3),
                    'steps': args.steps, 'step_ms': li_times.stats(),
                    'points_per_frame': int(lframes[0].size(0)), 'voxels_kept_per_gpu': int(lo.size(0)),
                    'window_sizes': sizes,
                    'roofline': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'kernel': 'sra_fwd_wave_k / sra_bwd_fused_k, events bound to every 5th launch of the timed steps',
                                 'sra_fwd': lstats('sra_fwd', SRA_BYTES_PER_TOKEN), 'sra_bwd': lstats('sra_bwd', SRA_BWD_BYTES_PER_TOKEN)}}

        lidar_leg = lidar_run(args.frames_per_gpu)
        lidar_leg['what'] = ('same step (fwd + bwd) on a LiDAR-like frame: 64 beams x 2650 azimuth steps over a ground plane, '
                             '5 % box hits, 5 % of the points outside the range (clamped), 2 % exact duplicates '
                             '(SURVEY.md section 8(d) L-cloud + pathological input); not part of `value`.  A sweep keeps 18 k voxels in '
                             '~800 windows of 23 tokens: one frame does not fill the chip (a launch of the attention core is bound by '
                             'the wave of its largest window, not by HBM) - the reference trains at 2 frames per GPU '
                             '(BASELINE.json configs[2]); `more_frames_per_gpu` = the same leg at 2 and 4')
        if args.frames_per_gpu == 1 and world == 1 and not os.environ.get('SST_BENCH_NO_LIDAR_BATCHES'):
            lidar_leg['more_frames_per_gpu'] = {}
            for nf in (2, 4):
                try:
                    r = lidar_run(nf)
                    for k_ in ('unit', 'steps', 'points_per_frame'):
                        r.pop(k_, None)
                    lidar_leg['more_frames_per_gpu'][str(nf)] = r
                except Exception as e:     # a side leg must never take the line down
                    lidar_leg['more_frames_per_gpu'][str(nf)] = {'error': repr(e)[:200]}

    # Beside the exact-fp32 headline: the same step with the projections / FFN products of the encoder layers evaluated as three
    # bf16 products of split fp32 operands with fp32 accumulation (csrc/dense_f32x3.hip; everything stays fp32 in HBM, the
    # attention core, LayerNorm and the weight gradients stay exact fp32).  ~1e-5 relative per product - tighter than the TF32
    # tensor-core products torch 1.8 (the reference's pinned version) uses for these layers by default on Ampere.
    def matmul_leg(mode, dtype, what):
        """the same step with the dense products of the encoder layers in another multiply mode (sst_amd/dense.py), beside
        `value`: W warm-up steps, K timed; its forward output against the timed mode's on the same frame"""
        note(f'leg: matmul mode {mode}')
        fresh_allocator()
        with torch.no_grad():
            ref_out, ref_key = gpu_forward_sorted(model, frames)
        model.backbone.set_precision(mode)
        try:
            with torch.no_grad():
                alt_out, alt_key = gpu_forward_sorted(model, frames)
            same = torch.equal(ref_key, alt_key)
            for _ in range(3):
                step()
            sync()
            mm_times = StepTimes()
            mm_times.mark()
            t4 = time.perf_counter()
            for _ in range(args.steps):
                step()
                mm_times.mark()
            sync()
            el = time.perf_counter() - t4
            mm_times.close()
        finally:
            model.backbone.set_precision('f32x6' if args.matmul == 'f32x6' else 'fp32')
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return {'value': round(world * args.frames_per_gpu * args.steps / el, 3), 'unit': 'frames/s',
                'ms_per_step': round(el / args.steps * 1e3, 3), 'steps': args.steps, 'dtype': dtype, 'step_ms': mm_times.stats(),
                'max_abs_err_vs_timed_mode_forward': float((alt_out - ref_out).abs().max()) if same else None,
                'voxels_equal': bool(same), 'what': what}

    x3_leg = mfma_leg = None
    if not args.fwd_only and not args.no_f32x3_leg and args.precision == 'f32':
        x3_leg = matmul_leg('f32x3', 'f32 storage, bf16 x 3 products',
                            'same step; q|k, v, out-proj, FFN products and their data gradients as x_hi w_hi + x_lo w_hi + x_hi w_lo '
                            'on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (two-way split: ~1e-5 relative per product, NOT '
                            'exact - a leg, never `value`); weight gradients on the fp32 matrix pipe')
        if args.matmul == 'f32x6':
            mfma_leg = matmul_leg('fp32', 'f32, fp32 matrix pipe',
                                  'same step with every dense product on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 / 32x32x2_f32: '
                                  'csrc/dense_f32.hip, csrc/wgrad.hip) - the arithmetic of rounds 1-3; the timed mode evaluates the '
                                  'same products from the exact three-way bf16 split (csrc/dense_f32x6.hip, csrc/wgrad_x6.hip)')

    # roofline of the dominant kernel group (SRA attention core, forward)
    def group_stats(kind):
        ev = [(e0.elapsed_time(e1), n) for k_, e0, e1, n in sink if k_ == kind]
        ev = [(t, n) for t, n in ev if t > 0]  # a launch that took another kernel path leaves its events unrecorded
        if not ev:
            return None
        ms = sum(t for t, _ in ev) / len(ev)
        tokens = sum(n for _, n in ev) / len(ev)
        return ms, tokens, len(ev)

    fwd = group_stats('sra_fwd')
    bwd = group_stats('sra_bwd')
    roofline = None
    if fwd is not None:
        ms, tokens, launches = fwd
        achieved = SRA_BYTES_PER_TOKEN * tokens / (ms * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': ('sra_fwd_wave_k<NTMAX, true> (scaled cosine attention; one launch per sst_sra_attn_cos_fwd_f32 call'
                                               if args.workload == 'sst_center' else
                                               'sra_fwd_wave_k<NTMAX, false> (one launch per sst_sra_attn_fwd_f32 call') +
                              '; HIP events bound to the launch, hipExtLaunchKernelGGL start/stop)',
                    'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
                    'algorithmic_bytes_per_launch': int(SRA_BYTES_PER_TOKEN * tokens),
                    'avg_launch_ms': round(ms, 4), 'launches_timed': launches,
                    'sampling': 'every 5th forward launch of the timed region'}
        if bwd is not None:
            # the other half of what `value` times: the one-pass backward kernel (dQ, dK, dV from one read of
            # Q, K, V, O, dO), same event method, and forward + backward of the attention core together
            bms, btokens, blaunches = bwd
            bach = SRA_BWD_BYTES_PER_TOKEN * btokens / (bms * 1e-3) / 1e9
            roofline['sra_bwd'] = {'kernel': 'sra_bwd_fused_k<NTMAX> (one launch per sst_sra_attn_bwd_f32 call)',
                                   'achieved': round(bach, 1), 'frac': round(bach / HBM_PEAK_GBS, 4), 'traffic': None,
                                   'algorithmic_bytes_per_launch': int(SRA_BWD_BYTES_PER_TOKEN * btokens),
                                   'avg_launch_ms': round(bms, 4), 'launches_timed': blaunches}
            both = (SRA_BYTES_PER_TOKEN * tokens + SRA_BWD_BYTES_PER_TOKEN * btokens) / ((ms + bms) * 1e-3) / 1e9
            roofline['sra_fwd_plus_bwd'] = {'achieved': round(both, 1), 'frac': round(both / HBM_PEAK_GBS, 4)}
        # HBM traffic of this kernel from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately by
        # tools/collect_sra_traffic.sh on the same workload and committed under profiles/): per launch, like `achieved`
        tpath = os.path.join(ROOT, 'profiles', 'latest_sra_traffic.json')
        source = 'profiles/latest_sra_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed)'
        live = remeasure_sra_traffic(args) if (rank == 0 and world == 1) else None
        if live is not None:
            tpath, source = live, ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/sra_only.py '
                                   '(tools/collect_sra_traffic.sh, gfx950 x2 FETCH correction)')
        if os.path.exists(tpath) and args.points == 116000 and args.frames_per_gpu == 1:
            try:
                tj = json.load(open(tpath))
                suffix = '_cosine' if args.workload == 'sst_center' else ''     # the <.., true> variants of the kernels
                roofline['traffic'] = int(tj['sra_fwd_wave_k' + suffix]['hbm_bytes_per_launch'])
                roofline['traffic_source'] = source
                if 'sra_bwd' in roofline and 'sra_bwd_fused_k' + suffix in tj:
                    roofline['sra_bwd']['traffic'] = int(tj['sra_bwd_fused_k' + suffix]['hbm_bytes_per_launch'])
                if bf16_leg is not None:
                    for key, kern in (('sra_fwd', 'sra_fwd_bf16_k'), ('sra_bwd', 'sra_bwd_bf16_k')):
                        if bf16_leg['roofline'].get(key) and kern in tj:
                            bf16_leg['roofline'][key]['traffic'] = int(tj[kern]['hbm_bytes_per_launch'])
            except Exception:
                pass

    if rank == 0:
        total_frames = world * args.frames_per_gpu * args.steps
        res = {
            'metric': 'LiDAR frames/sec (SST backbone fwd+bwd) at Waymo 0.32m voxels' if not args.fwd_only
            else 'LiDAR frames/sec (SST backbone fwd-only) at Waymo 0.32m voxels',
            'value': round(total_frames / elapsed, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'step_ms': main_times.stats(),
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'host_ms_per_step': main_host_ms,   # interpreter + launch time of a step (forward and backward timed with the device
                                                # idle at their start): the step is bound by the kernels while this stays below ms_per_step
            'dtype': ('f32 (storage, accumulation, attention, LayerNorm, reductions: fp32; the dense products of the encoder '
                      'layers from the EXACT three-way bf16 split of both fp32 operands, six bf16 MFMA products with fp32 '
                      'accumulation - error vs float64 <= 2 x the fp32 matrix pipe\'s: tests/test_gpu_dense_f32x6.py)'
                      if (args.precision == 'f32' and args.matmul == 'f32x6') else args.precision),
            'data': 'synthetic',
            'gemm_tuning': 'none (no library GEMM in this step)',
            'config': {'workload': ('NOT THE HEADLINE WORKLOAD (--cloud lidar): LiDAR-like synthetic sweep, ' if args.cloud == 'lidar'
                                    else 'SST-base Waymo training, bs=2/GPU, 0.32 m voxel: uniform synthetic cloud '
                                    if args.workload == 'sst_bs2' else
                                    'NOT THE HEADLINE WORKLOAD (--workload sst_center): configs/sst_refactor/'
                                    'sst_waymoD5_1x_3class_centerhead.py as shipped (cosine attention, checkpoint_blocks=[0, 1], '
                                    '5-channel points), voxel features scope: uniform synthetic cloud '
                                    if args.workload == 'sst_center' else
                                    'SST-base Waymo single-frame, 0.32 m voxel: uniform synthetic cloud ') +
                                   f'{args.points} points/frame -> {n_voxels // args.frames_per_gpu} non-empty '
                                   'voxels/frame; dynamic voxelize + DynamicVFE + SSTInputLayerV2 + '
                                   f'{args.blocks} SRA blocks' + (' + BEV canvas + 3 attached convolutions' if args.workload == 'sst_bev' else '') + ', '
                                   + ('fwd only' if args.fwd_only else 'fwd+bwd'),
                       'frames_per_gpu': args.frames_per_gpu, 'points_per_frame': args.points,
                       'voxels_per_gpu': n_voxels, 'parallelism': f'dp{world}',
                       'index_plan': 'built at the head of its step' if args.no_plan_prefetch else
                                     ('one plan per step, built on the planner\'s own stream beside the previous step\'s backward '
                                      'pass (DynamicVoxelNet.prepare(.., overlap=True))' if args.plan_overlap else
                                      'one plan per step, queued behind the previous step\'s backward pass (same stream)'),
                       'grad_sync': ('one flat persistent fp32 buffer, ' + (f'{len(reducer.buckets)} bucket(s) sent from autograd '
                                     'hooks during the backward pass' if args.grad_sync == 'overlap' else
                                     'one blocking all-reduce after the backward pass') + ', over '
                                     + ('RCCL' if args.backend == 'nccl' else args.backend)) if world > 1 else 'none'},
            'roofline': roofline,
        }
        if comm is not None:
            # the exchanges of one step when nothing overlaps them; in the timed step all but the last bucket ride under the
            # backward pass of the voxel encoder / index stages
            res.update(allreduce_ms=comm['allreduce_ms'], bn_sync_ms=comm['bn_sync_ms'], communication=comm)
        if world > 1:
            try:
                lib_version = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                lib_version = None
            res['ranks'] = {'world_size': world, 'backend': args.backend, 'collective_library':
                            ('RCCL ' + lib_version) if (args.backend == 'nccl' and lib_version) else args.backend,
                            'reduce_op_avg': bool(reducer is not None and reducer._avg), 'per_rank': per_rank,
                            'devices_visible': torch.cuda.device_count()}
        if as_is is not None:
            as_is['vs_value'] = round(as_is['value'] / res['value'], 4)
            res['config_as_is'] = as_is
        if fwd_only is not None:
            res['forward_only'] = fwd_only
        if lidar_leg is not None:
            res['lidar_like_cloud'] = lidar_leg
        if x3_leg is not None:
            res['precision_f32x3'] = x3_leg
        if mfma_leg is not None:
            res['precision_f32_mfma'] = mfma_leg
        if bf16_leg is not None:
            res['reduced_precision'] = bf16_leg
        if world == 1 and not args.no_cpu_baseline:
            note('leg: cpu_baseline + parity')
            res['cpu_baseline'], res['parity'] = cpu_reference_leg(model, frames[0].cpu(), args.blocks)
        else:
            res['cpu_baseline'] = None
        if std_leg is not None:
            res['std_attention_same_config'] = std_leg
        if world == 1 and not args.no_voxelize_roofline:
            note('leg: voxelize roofline')
            try:
                rv = voxelize_roofline(dev, args.points)
            except Exception as e:
                rv = {'error': repr(e)[:200]}
            if isinstance(res.get('roofline'), dict):
                res['roofline']['voxelize'] = rv
            else:
                res['roofline_voxelize'] = rv
        if world == 1 and args.workload == 'sst' and not args.no_workload_legs and not args.fwd_only:
            del model
            fresh_allocator()
            res['workloads'] = workload_legs(args)
        print(json.dumps(res), file=line_out)
        line_out.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
