#!/usr/bin/env python
"""Static check of the hand-pipelined kernels (wgrad.hip, tall_gemm.hip): registers that are the destination of an
inline-asm global_load (data arrives asynchronously, outside the compiler's knowledge) must only ever be touched by
inline-asm statements (waits, MFMAs, refills) — a compiler-generated copy / spill / VALU read of such a register
could observe it before the data has landed.  Usage: check_async_regs.py <file.s> <kernel-name-substring>"""
import re
import sys


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def main(path, name):
    lines = open(path).read().split('\n')
    bad = 0
    for k0, ln in enumerate(lines):
        if re.match(r'^_Z\w*' + re.escape(name) + r'\w*:', ln):
            in_asm = False
            async_regs = set()
            body = []
            for l in lines[k0:]:
                if 's_endpgm' in l:
                    break
                body.append(l)
            for l in body:
                if '#ASMSTART' in l:
                    in_asm = True
                elif '#ASMEND' in l:
                    in_asm = False
                elif in_asm and 'global_load' in l:
                    async_regs |= regs(l.split()[1].rstrip(','))
            in_asm = False
            n_bad = 0
            for l in body:
                if '#ASMSTART' in l:
                    in_asm = True
                    continue
                if '#ASMEND' in l:
                    in_asm = False
                    continue
                t = l.strip()
                if in_asm or not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
                    continue
                toks = re.split(r'[ ,]+', t.split(';')[0].strip())
                used = set()
                for tok in toks[1:]:
                    used |= regs(tok)
                if used & async_regs and not toks[0].startswith('s_'):
                    # allowed: touching them before the first asm load / after the final wait is fine, but we cannot
                    # tell statically; report everything and let the reader judge
                    n_bad += 1
                    if n_bad <= 12:
                        print('   ', t[:100])
            print(f'{ln[:70]}  async regs: {len(async_regs)}  compiler instructions touching them: {n_bad}')
            bad += n_bad
    return bad


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
