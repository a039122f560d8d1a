#!/usr/bin/env python
"""Build container only (needs /root/reference): the stub-loaded REFERENCE pipeline (its compiled dynamic_voxelize ->
its DynamicVFE -> SSTInputLayerV2 -> SSTv2, tests/test_ref_assembly.ReferenceAssembly) timed beside the CPU port
(oracle/cpu_pipeline.CpuSSTBackbone = bench.py's cpu_baseline, kind "port") on the SAME host cores, same clouds, same
weights: the measured ratio port / reference that stands behind `cpu_baseline.kind: "port"`.
Protocol of SURVEY.md §8(d): warm-up passes, then timed passes, median; thread counts 1 and all.
Usage: python tools/cpu_ref_vs_port.py [--full] profiles/rNN/cpu_ref_vs_port.json   (the reference prints to stdout)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_ref_assembly as A  # noqa: E402


def timed(fn, warm, reps):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def case(name, clouds, blocks, backward, warm, reps, threads):
    ref, port = A.build_pair(blocks)
    res = {'case': name, 'points': [int(c.size(0)) for c in clouds], 'blocks': blocks,
           'pass': 'forward + backward' if backward else 'forward', 'warmup': warm, 'timed': reps, 'threads': {}}

    def run(net, is_ref):
        def f():
            if backward:
                net.zero_grad(set_to_none=True)
                out = net(clouds)
                out = out[0] if is_ref else out
                out.sum().backward()
            else:
                with torch.no_grad():
                    net(clouds)
        return f

    for t in threads:
        torch.set_num_threads(t)
        r, rs = timed(run(ref, True), warm, reps)
        p, ps = timed(run(port, False), warm, reps)
        res['threads'][str(t)] = {'reference_s': round(r, 4), 'port_s': round(p, 4), 'port_over_reference': round(p / r, 3),
                                  'reference_passes_s': [round(v, 3) for v in rs], 'port_passes_s': [round(v, 3) for v in ps]}
    with torch.no_grad():
        want, wc = ref(clouds)
        got = port(clouds)
    res['max_abs_diff'] = float((want - got).abs().max())
    res['voxels'] = int(got.size(0))
    res['voxel_rows_equal'] = bool(np.array_equal(port.last_voxel_coors.numpy().astype(np.int64), wc.numpy().astype(np.int64)))
    return res


def main():
    full = '--full' in sys.argv
    all_threads = os.cpu_count()
    out = {'host_cores': all_threads, 'torch': torch.__version__,
           'what': 'reference (stub-loaded, unmodified Python + compiled dynamic_voxelize) vs the CPU port of the same data '
                   'flow, same cores / clouds / weights; DynamicScatter and the in-window rank are the oracle restatements on '
                   'both sides (GPU-only / un-vendored in the reference)',
           'cases': []}
    out['cases'].append(case('BASELINE configs[0]: 20k points, voxelize + DynamicVFE + 1 SRA block', [A.uniform_cloud(20000, 7)],
                             1, False, 3, 10, [1, all_threads]))
    out['cases'].append(case('20k points, 1 SRA block', [A.uniform_cloud(20000, 7)], 1, True, 2, 5, [1, all_threads]))
    if full:
        out['cases'].append(case('headline frame: 116k points, 6 SRA blocks', [A.uniform_cloud(116000, 0)], 6, True, 1, 2,
                                 [all_threads]))
    dst = [a for a in sys.argv[1:] if not a.startswith('--')]
    text = json.dumps(out, indent=1)
    if dst:
        open(dst[0], 'w').write(text + '\n')
    else:
        print(text)


if __name__ == '__main__':
    main()
