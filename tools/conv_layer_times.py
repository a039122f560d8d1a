"""Per-layer durations of the sparse-convolution kernels inside one run of `tools/microbench.py unet` from a rocprofv3
kernel trace: groups the dispatches by (kernel, grid) -- the grid identifies the U-Net level and the column groups.
Usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pu -o u -- python tools/microbench.py unet
                 python tools/conv_layer_times.py /tmp/pu/u_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    for key in ('sp_conv_seg_k', 'sp_gather_gemm_k', 'sp_wgrad_k'):
        if key in n:
            gx = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
            gy = int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))
            a = agg.setdefault((key, gx, gy), [0, 0])
            a[0] += 1
            a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f'{k[0]:18s} grid {k[1]:6d} x {k[2]:3d}: {c:5d} calls, avg {t / c / 1000:8.1f} us, total {t / 1e6:8.2f} ms')
