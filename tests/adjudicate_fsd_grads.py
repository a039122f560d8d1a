"""TEST INFRASTRUCTURE — adjudication of the FSD / FSDv2 parameter-gradient differences at full size (VERDICT round 3, item 2:
"3.9e-3 - 5.3e-3 against the CPU port; nothing adjudicates with a float64 evaluation of the port").

Three evaluations of the SAME chain (bench_workloads.FSDPath / FSDv2Path), same weights, same frame, training mode:
  gpu     sst_amd, fp32 (the product)                                         -> `gpu` stage, on the GPU box
  port32  oracle/fsd_cpu.py, fp32 (the reference's algorithm; = cpu_baseline) -> `cpu` stage, anywhere
  port64  oracle/fsd_cpu.py, float64 (integer stages - voxelisation, foreground selection, clustering - in fp32, so that
          the decisions upstream are those of the fp32 runs): the adjudicator
and, between port32 and port64, the DISCRETE decisions of the forward pass that an fp32 rounding difference can flip: the
sign of every ReLU input and the arg-max row of every segmented maximum (voxel encoder / SIR pooling).  A flipped decision
moves a gradient contribution from one row to another: the gradient is a discontinuous function of the activations there,
and two correct fp32 evaluations differ by O(flips / rows), not by O(eps).

  python tests/adjudicate_fsd_grads.py gpu --workload fsd [--points N] --out gpurun_out/adj_fsd.pt
  python tests/adjudicate_fsd_grads.py cpu --in gpurun_out/adj_fsd.pt --out profiles/r04/fsd_grad_adjudication.json

The gpu stage stores gradients only (weights and cloud are re-created from the seeds; a checksum guards it); parameters
above 20 k elements are stored as a strided subsample (the statistic is a maximum: a subsample bounds it from below and is
compared like for like).  tests/test_fsd_chain.py::test_gpu_gradients_within_fp32_noise_at_40k runs the same functions."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SUBSAMPLE_ABOVE = 20_000


def _sub(t):
    t = t.detach().reshape(-1)
    if t.numel() > SUBSAMPLE_ABOVE:
        t = t[::-(-t.numel() // SUBSAMPLE_ABOVE)]
    return t


def build(workload, ops, points, roi_stage=True, seed=0):
    import bench_workloads as BW
    spec = BW.WORKLOADS[workload]
    torch.manual_seed(seed)
    kw = dict(roi_stage=roi_stage) if workload == 'fsd' else {}
    net = spec['cls'](ops, **kw).train()
    cloud = net.make_cloud(points, 0, 'cpu')
    return net, cloud


def checksum(state):
    return float(sum(v.double().abs().sum() for v in state.values() if v.is_floating_point()))


# ---------------------------------------------------------------------------------------------------------------- decisions
class Decisions(object):
    """records (first run) / compares (second run) the discrete decisions of a forward pass of the CPU port: ReLU input signs
    (torch.nn.functional.relu, which nn.ReLU calls) and the arg-max rows of fsd_cpu._segment(..., 'max')"""

    def __init__(self):
        self.sites, self.mode, self.at = [], 'record', 0
        self.relu_total = self.relu_flips = self.max_total = self.max_flips = 0
        self.per_site = []

    def _push(self, kind, packed, n):
        if self.mode == 'record':
            self.sites.append((kind, packed, n))
            return
        k0, p0, n0 = self.sites[self.at]
        self.at += 1
        if k0 != kind or n0 != n:
            self.per_site.append((kind, n, None))            # shapes differ: an integer stage upstream flipped
            return
        if kind == 'relu':
            flips = int(np.unpackbits(np.bitwise_xor(p0, packed)).sum())
            self.relu_total += n
            self.relu_flips += flips
        else:
            flips = int((p0 != packed).sum())
            self.max_total += n
            self.max_flips += flips
        self.per_site.append((kind, n, flips))

    def relu(self, x):
        self._push('relu', np.packbits((x.detach() > 0).reshape(-1).numpy()), x.numel())

    def segmax(self, src, index, out):
        # smallest source row that attains the maximum of its (group, channel): the row the gradient goes to
        hit = src.detach() == out.detach()[index]
        rows = torch.arange(src.size(0)).view(-1, 1).expand_as(src)
        big = torch.full_like(rows, src.size(0))
        arg = torch.full(out.shape, src.size(0), dtype=torch.long).scatter_reduce(
            0, index.view(-1, 1).expand_as(src), torch.where(hit, rows, big), reduce='amin', include_self=True)
        self._push('max', arg.reshape(-1).numpy().astype(np.int32), arg.numel())

    def __enter__(self):
        import torch.nn.functional as F
        from oracle import fsd_cpu
        self._relu, self._seg = F.relu, fsd_cpu._segment
        me = self

        def relu(x, inplace=False):
            me.relu(x)
            return me._relu(x, inplace=inplace)

        def segment(src, index, n_out, reduce):
            out = me._seg(src, index, n_out, reduce)
            if reduce == 'max':
                me.segmax(src, index, out)
            return out

        F.relu, fsd_cpu._segment = relu, segment
        torch.relu_backup = None
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        from oracle import fsd_cpu
        F.relu, fsd_cpu._segment = self._relu, self._seg
        return False


def run_port(workload, points, dtype, decisions=None, roi_stage=True, state=None, cloud=None, perturb_ulp=0.0):
    """-> (gradients {name: flat float64 tensor, subsampled}, stats, seconds)"""
    from oracle import fsd_cpu
    net, made = build(workload, fsd_cpu, points, roi_stage)
    cloud = made if cloud is None else cloud
    if state is not None:
        net.load_state_dict(state, strict=True)
    ck = checksum(net.state_dict())
    if dtype == torch.float64:
        net = net.double()
    if perturb_ulp:
        # every weight moved by a random fraction of ONE fp32 rounding step (2^-24 relative): what an fp32 evaluation cannot
        # distinguish from the weights it was given
        g = torch.Generator().manual_seed(123)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.0 + perturb_ulp * 2.0 ** -24 * (torch.rand(p.shape, generator=g, dtype=torch.float64) * 2 - 1).to(p.dtype))
    t0 = time.perf_counter()
    if decisions is not None:
        with decisions:
            loss, stats = net([cloud.to(dtype)])
    else:
        loss, stats = net([cloud.to(dtype)])
    loss.backward()
    secs = time.perf_counter() - t0
    grads = {n: _sub(p.grad).double().clone() for n, p in net.named_parameters() if p.grad is not None}
    return grads, {k: int(v) for k, v in stats.items()}, secs, ck


def compare(a, b):
    """max |a - b| / max |b| per parameter"""
    return {n: float((a[n] - b[n]).abs().max() / b[n].abs().max().clamp(min=1e-30)) for n in b if n in a}


def adjudicate(workload, points, gpu_grads=None, gpu_stats=None, roi_stage=True, log=print, sensitivity=True):
    dec = Decisions()
    g32, s32, t32, ck = run_port(workload, points, torch.float32, dec, roi_stage)
    log(f'port32: {t32:.1f} s, {s32}')
    dec.mode = 'compare'
    g64, s64, t64, _ = run_port(workload, points, torch.float64, dec, roi_stage)
    log(f'port64: {t64:.1f} s, {s64}')
    res = {'workload': workload, 'points': points, 'roi_stage': roi_stage, 'sizes': s32, 'integer_stages_equal_port32_port64': s32 == s64,
           'weights_checksum': ck, 'seconds': {'port32': round(t32, 1), 'port64': round(t64, 1)},
           'decisions_port32_vs_port64': {
               'relu_inputs': dec.relu_total, 'relu_sign_flips': dec.relu_flips,
               'segmented_max_outputs': dec.max_total, 'argmax_row_flips': dec.max_flips,
               'sites_with_flips': sum(1 for _, _, f in dec.per_site if f), 'sites': len(dec.per_site),
               'what': 'discrete decisions of ONE forward pass that differ between the fp32 and the float64 evaluation of the '
                       'SAME CPU algorithm on the same weights and frame (sign of every ReLU input; source row of every '
                       'segmented maximum)'}}
    p32 = compare(g32, g64)
    rows = {n: {'port32_vs_f64': p32[n]} for n in p32}
    if sensitivity:
        # the float64 evaluation again, every weight moved by a random fraction of one fp32 rounding step: how far the exact
        # gradient moves under a perturbation no fp32 evaluation can see (an fp32 run makes such an error at EVERY operation)
        g64p, s64p, _, _ = run_port(workload, points, torch.float64, None, roi_stage, perturb_ulp=1.0)
        if s64p == s64:
            for n, v in compare(g64p, g64).items():
                rows[n]['f64_moved_by_one_fp32_ulp_of_the_weights'] = v
    if gpu_grads is not None:
        gg = {n: v.double() for n, v in gpu_grads.items()}
        gv64, gv32 = compare(gg, g64), compare(gg, g32)
        for n in rows:
            if n in gv64:
                rows[n].update(gpu_vs_f64=gv64[n], gpu_vs_port32=gv32[n])
        res['integer_stages_equal_gpu_port32'] = gpu_stats == s32 if gpu_stats is not None else None
    worst = sorted(rows.items(), key=lambda kv: -kv[1]['port32_vs_f64'])
    res['parameters'] = len(rows)
    res['max_over_parameters'] = {k: max(r[k] for r in rows.values() if k in r) for k in ('port32_vs_f64', 'f64_moved_by_one_fp32_ulp_of_the_weights', 'gpu_vs_f64', 'gpu_vs_port32')
                                  if any(k in r for r in rows.values())}
    if gpu_grads is not None:
        ratio = [r['gpu_vs_f64'] / max(r['port32_vs_f64'], 1e-9) for r in rows.values() if 'gpu_vs_f64' in r]
        res['gpu_error_over_port32_error'] = {'median': float(np.median(ratio)), 'max': float(np.max(ratio)),
                                              'parameters_where_gpu_is_closer_to_f64_than_port32': int(sum(1 for x in ratio if x <= 1.0))}
        res['parameters_above_1e-3_vs_f64'] = {'gpu': sum(1 for r in rows.values() if r.get('gpu_vs_f64', 0) > 1e-3),
                                               'port32': sum(1 for r in rows.values() if r['port32_vs_f64'] > 1e-3)}
    res['rows'] = rows
    res['worst_parameters'] = [{'name': n, **{k: float(f'{v:.3e}') for k, v in r.items()}} for n, r in worst[:12]]
    for key in ('seg_backbone.conv_input.0.weight', 'seg_head.weight', 'backbone.block_list.0.vfe_layers.0.linear.weight',
                'virtual_stage.backbone.conv_out.0.weight'):
        if key in rows:
            res.setdefault('bench_line_parameters', {})[key] = {k: float(f'{v:.3e}') for k, v in rows[key].items()}
    return res


def gpu_stage(args):
    import bench_workloads as BW
    dev = torch.device('cuda:0')
    net, cloud = build(args.workload, BW.GpuOps, args.points, not args.no_roi)
    ck = checksum(net.state_dict())
    net = net.to(dev)
    loss, stats = net([cloud.to(dev)])
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: _sub(p.grad).cpu() for n, p in net.named_parameters() if p.grad is not None}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    torch.save({'workload': args.workload, 'points': args.points, 'roi_stage': not args.no_roi, 'checksum': ck,
                'stats': {k: int(v) for k, v in stats.items()}, 'grads': grads}, args.out)
    print(json.dumps({'saved': args.out, 'parameters': len(grads), 'stats': {k: int(v) for k, v in stats.items()},
                      'mbytes': round(sum(g.numel() for g in grads.values()) * 4 / 1e6, 1)}))


def cpu_stage(args):
    blob = torch.load(args.inp) if args.inp else None
    workload = blob['workload'] if blob else args.workload
    points = blob['points'] if blob else args.points
    roi = blob['roi_stage'] if blob else not args.no_roi
    res = adjudicate(workload, points, blob['grads'] if blob else None, blob['stats'] if blob else None, roi)
    if blob is not None:
        res['weights_checksum_matches_gpu_box'] = abs(res['weights_checksum'] - blob['checksum']) <= 1e-6 * abs(blob['checksum'])
    res['threads'] = torch.get_num_threads()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    rows = res.pop('rows')
    res['per_parameter'] = {n: {k: float(f'{v:.3e}') for k, v in r.items()} for n, r in rows.items()}
    if args.out:
        json.dump(res, open(args.out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ('worst_parameters', 'per_parameter')}, indent=1))
    for row in res['worst_parameters']:
        print(row)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('stage', choices=['gpu', 'cpu'])
    ap.add_argument('--workload', default='fsd', choices=['fsd', 'fsdv2'])
    ap.add_argument('--points', type=int, default=None)
    ap.add_argument('--no-roi', action='store_true')
    ap.add_argument('--in', dest='inp', default=None)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    if a.points is None:
        a.points = 160000 if a.workload == 'fsd' else 300000
    (gpu_stage if a.stage == 'gpu' else cpu_stage)(a)
