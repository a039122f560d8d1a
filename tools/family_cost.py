#!/usr/bin/env python
"""What every kernel family costs a launch-bound step: duration PLUS the idle gap behind each launch (until the next
kernel starts), per family, from a rocprofv3 --kernel-trace CSV.  A 5 us kernel followed by 4 us of dispatch gap costs
9 us of wall time; families with many short launches show up here, not in the duration statistics.
Usage: family_cost.py <kernel_trace.csv[.gz]> [steps] [skip_fraction]"""
import csv
import gzip
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    m = re.match(r'(Cijk_\w+?_MT\d+x\d+x\d+)', n)
    if m:
        return 'hipBLASLt ' + m.group(1)
    return re.sub(r'\(.*', '', n)[:64]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    f = gzip.open(path, 'rt') if path.endswith('.gz') else open(path)
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(f))
    rows = rows[int(len(rows) * skip):]
    fam = {}
    for i, (s, e, n) in enumerate(rows):
        gap = max(0, rows[i + 1][0] - e) if i + 1 < len(rows) else 0
        gap = min(gap, 50000)        # host stalls (read-backs) are not the kernel's cost
        t = fam.setdefault(short(n), [0, 0, 0])
        t[0] += e - s
        t[1] += gap
        t[2] += 1
    span = rows[-1][1] - rows[0][0]
    print(f'{len(rows)} launches, span {span / 1e6:.2f} ms = {span / 1e6 / steps:.2f} ms/step over {steps:g} steps')
    print(f'{"ms/step":>9s} {"kernel":>8s} {"gap":>8s} {"calls/step":>10s}  family')
    for k, (d, g, c) in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
        print(f'{(d + g) / 1e6 / steps:9.3f} {d / 1e6 / steps:8.3f} {g / 1e6 / steps:8.3f} {c / steps:10.1f}  {k}')


if __name__ == '__main__':
    main()
