#!/usr/bin/env python
"""Summary of tools/collect_conv_pmc.sh: per layer, the counters of the LAST five launches of sp_conv_os_k (the loop of
tools/conv_only.py, not the capture pass), and what they say: MFMA pipe busy share, CUs busy share of the launch.
Usage: conv_pmc_summary.py <dir> <layer> [<layer> ...]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    for layer in sys.argv[2:]:
        vals, meta = {}, None
        for f in sorted(glob.glob(f'{d}/L{layer}_set*.csv')):
            rows = list(csv.DictReader(open(f)))
            if not rows:
                continue
            ids = sorted({int(r['Dispatch_Id']) for r in rows})[-5:]
            for r in rows:
                if int(r['Dispatch_Id']) in ids:
                    vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
                    meta = (r['Grid_Size'], r['Workgroup_Size'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        if not vals:
            print(f'layer {layer}: no counters')
            continue
        avg = {k: sum(v) / len(v) for k, v in vals.items()}
        print(f'layer {layer}: grid {meta[0]} x {meta[1]}, last launch {meta[2]:.0f} us (under the counters)')
        for k in sorted(avg):
            print(f'   {k:28s} {avg[k]:16.0f}')
        gui = avg.get('GRBM_GUI_ACTIVE')
        if gui:
            cyc = gui / 8.0     # the counter sums the 8 XCDs
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
                print(f'   -> MFMA pipe busy {100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc):.1f} % of the SIMD-cycles of the launch '
                      f'({avg.get("SQ_INSTS_MFMA", 0) / 1e6:.2f} M MFMAs x 32 cycles)')
            if 'SQ_BUSY_CU_CYCLES' in avg:
                print(f'   -> CUs busy {100 * avg["SQ_BUSY_CU_CYCLES"] / (256 * cyc):.1f} % of the launch')


if __name__ == '__main__':
    main()
