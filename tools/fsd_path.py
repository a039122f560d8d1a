"""FSD's hot path end to end on synthetic data, with the modules of this repository wired as the reference wires them
(detector glue -- heads, losses, box decoding -- replaced by small linear layers; detectors are out of scope):

  points -> Voxelization -> DynamicScatterVFE -> SimpleSparseUNet          VoteSegmentor.extract_feat, single_stage_fsd.py:228-250
         -> voxel-to-point gather + local xyz                             Voxel2PointScatterNeck, necks/voxel2point_neck.py:28-63
         -> (linear stand-in for the vote / segmentation head)
         -> foreground points, voted centres -> ClusterAssigner           single_stage_fsd.py:922-999 (connected components)
         -> scatter_v2 cluster centres, SIR x 3                            SingleStageFSD.extract_feat, :467-483
         -> (linear stand-in for the box head) -> RoIs
         -> DynamicPointROIExtractor -> SIRLayer x 2 on the pooled points  roi_heads/.../dynamic_point_roi_extractor.py, fsd_bbox_head.py:69-97

Prints forward and forward + backward time per frame.  `python tools/fsd_path.py [points_per_frame] [frames]`."""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads  # noqa: E402  (the pipeline itself lives beside bench.py: `python bench.py --workload fsd`)
from bench_workloads import FSDPath  # noqa: E402

DEV = 'cuda:0'


def lidar_like_cloud(n, seed):
    return bench_workloads.lidar_like_cloud(n, seed, DEV)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    torch.manual_seed(0)
    net = FSDPath().to(DEV).train()
    clouds = [lidar_like_cloud(n, i)[0] for i in range(frames)]

    def step(backward):
        for p in net.parameters():
            p.grad = None
        loss, stats = net(clouds)
        if backward:
            loss.backward()
        return stats

    for _ in range(3):
        stats = step(True)
    times = {}
    for name, bw in (('forward', False), ('forward + backward', True)):
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if bw:
                step(True)
            else:
                with torch.no_grad():
                    step(False)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        times[name] = float(np.median(ts))
    print(f'FSD path, {frames} frame(s) x {n} points: {stats}; forward {times["forward"]:.1f} ms, '
          f'forward + backward {times["forward + backward"]:.1f} ms '
          f'({frames * 1000.0 / times["forward + backward"]:.1f} frames/s fwd+bwd, {frames * 1000.0 / times["forward"]:.1f} fwd)')


if __name__ == '__main__':
    main()
