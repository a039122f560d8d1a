#!/usr/bin/env python
"""Where a workload synchronises the host with the device (GPU box): one forward + backward under
torch.cuda.set_sync_debug_mode('warn'), call sites counted.  Usage: python tools/sync_sites.py fsd|fsdv2"""
import collections
import os
import sys
import traceback
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as BW  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'fsd'
    spec = BW.WORKLOADS[what]
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = spec['cls']().to(dev).train()
    clouds = [model.make_cloud(spec['points'], 0, dev)]
    for _ in range(2):
        loss, _ = model(clouds)
        loss.backward()
    torch.cuda.synchronize()
    sites = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def hook(message, category, filename, lineno, file=None, line=None):
        if 'synchroniz' not in str(message):
            return
        stack = [f for f in traceback.extract_stack() if f.filename.startswith(root) and 'sync_sites' not in f.filename]
        key = ' <- '.join(f'{os.path.relpath(f.filename, root)}:{f.lineno}' for f in reversed(stack[-3:]))
        sites[key] += 1

    warnings.showwarning = hook
    warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode('warn')
    loss, _ = model(clouds)
    n_fwd = sum(sites.values())
    loss.backward()
    torch.cuda.set_sync_debug_mode('default')
    print(f'{what}: {sum(sites.values())} host synchronisations per step ({n_fwd} in the forward pass)')
    for k, v in sites.most_common(60):
        print(f'{v:4d}  {k}')


if __name__ == '__main__':
    main()
