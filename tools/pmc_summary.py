#!/usr/bin/env python
"""Averages rocprofv3 --pmc *_counter_collection.csv per kernel (kernel names matching a regex)."""
import collections
import csv
import re
import sys


def main(paths, pat):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            m = re.search(pat, r['Kernel_Name'])
            if not m:
                continue
            acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(acc):
        print(k)
        for c, vals in sorted(acc[k].items()):
            print(f'   {c:32s} {sum(vals) / len(vals):16.1f}   (n={len(vals)})')


if __name__ == '__main__':
    main(sys.argv[2:], sys.argv[1])
