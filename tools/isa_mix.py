#!/usr/bin/env python
"""Instruction mix of the hottest loop (the basic-block run with the most MFMAs between a label and its back branch) of every
kernel in a hipcc -S listing whose name matches a regex:  python tools/isa_mix.py file.s 'v9_kernelILi2ELi1ELi4'"""
import re
import sys


def classify(l):
    op = l.split()[0]
    if 'mfma' in op: return 'mfma'
    if 'buffer_load' in op and ' lds' in l: return 'lds_dma'
    if 'buffer_load' in op or 'global_load' in op: return 'vmem_load'
    if 'buffer_store' in op or 'global_store' in op: return 'vmem_store'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'ds_read'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'ds_write'
    if op.startswith('scratch_'): return 'scratch'
    if op == 's_waitcnt': return 'waitcnt'
    if op == 's_barrier': return 'barrier'
    if op == 's_nop': return 'nop'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main(path, pat):
    text = open(path).read().split('\n')
    starts = [i for i, l in enumerate(text) if re.match(r'^_Z\w+:', l)]
    for si, st in enumerate(starts):
        name = text[st].split(':')[0]
        if not re.search(pat, name): continue
        end = starts[si + 1] if si + 1 < len(starts) else len(text)
        body = text[st:end]
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r'^(\.LBB\w+):', l)
            if m: labels[m.group(1)] = i
        best = None
        for i, l in enumerate(body):
            m = re.search(r's_cbranch_\w+\s+(\.LBB\w+)', l) or re.search(r's_branch\s+(\.LBB\w+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = [x.strip() for x in body[labels[m.group(1)]:i + 1] if x.strip() and not x.strip().startswith(('.', ';'))]
                n = sum('mfma' in x.split()[0] for x in seg)
                if best is None or n > best[0]: best = (n, seg)
        if not best: continue
        cnt = {}
        for l in best[1]:
            k = classify(l); cnt[k] = cnt.get(k, 0) + 1
        print(name[:110]); print('   loop:', dict(sorted(cnt.items())))
        waits = [l for l in best[1] if l.startswith('s_waitcnt')]
        print('   waits:', ' | '.join(w.replace('s_waitcnt ', '') for w in waits))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '.')
