#!/usr/bin/env python
"""Lists the kernels of ONE steady-state step between the voxelize launch and the first SRA forward launch (the index
phase + VFE) from a rocprofv3 --kernel-trace CSV, in launch order, with start offsets, durations and the gaps.
Usage: front_of_step.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])))
rows = rows[int(len(rows) * 0.7):]
a = next(i for i, r in enumerate(rows) if 'dynamic_voxelize_' in r[2])
b = next(i for i in range(a, len(rows)) if 'sra_fwd' in rows[i][2])
t0, prev_end = rows[a][0], rows[a][0]
for s, e, n in rows[a:b + 1]:
    n = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n)[:78]
    print(f'{(s - t0) / 1e3:8.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:6.1f} us  {n}')
    prev_end = max(prev_end, e)
print(f'{b - a} launches, span {(rows[b][0] - t0) / 1e3:.1f} us')
