#!/usr/bin/env python
"""Compact view of a rocprofv3 *_kernel_stats.csv: ms per step and calls per step for the top kernels."""
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'(Cijk_\w+?_MT\d+x\d+x\d+)', name)
    if m:
        return 'hipBLASLt ' + m.group(1)
    name = re.sub(r'at::native::', '', name)
    return name[:90]


def main(path, steps, top=30):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r['TotalDurationNs']) for r in rows)
    print(f'total kernel time {tot / 1e6 / steps:.3f} ms/step over {steps} steps; {len(rows)} distinct kernels')
    for r in rows[:top]:
        print(f"{int(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step  {int(r['Calls']) / steps:7.1f} calls/step  "
              f"avg {float(r['AverageNs']) / 1e3:8.1f} us  {short(r['Name'])}")


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 30)
