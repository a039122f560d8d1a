#!/usr/bin/env python
"""Static instruction mix of one kernel in a hipcc -S output: python tools/asm_mix.py file.s <name-substring>"""
import collections
import re
import sys


def mix(path, pat):
    txt = open(path).read()
    out = []
    for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):.*?\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
        ops = collections.Counter()
        for line in m.group(2).split('\n'):
            line = line.strip()
            if not line or line.startswith((';', '.')) or line.endswith(':'):
                continue
            ops[line.split()[0]] += 1
        cls = collections.Counter()
        for op, n in ops.items():
            if op.startswith('v_mfma'):
                cls['mfma'] += n
            elif op.startswith('v_'):
                cls['valu'] += n
            elif op.startswith('s_'):
                cls['salu'] += n
            elif op.startswith(('global_', 'buffer_', 'flat_')):
                cls['vmem'] += n
            elif op.startswith('ds_'):
                cls['ds'] += n
        out.append((m.group(1), sum(ops.values()), dict(cls), ops))
    return out


if __name__ == '__main__':
    for name, tot, cls, ops in mix(sys.argv[1], sys.argv[2]):
        print(name[:60], 'total', tot, cls)
        print('   ', ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 16))
