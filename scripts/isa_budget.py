#!/usr/bin/env python3
"""Instruction budget of a kernel from its assembly (hipcc --save-temps: *.s): per basic block, by category.

    python scripts/isa_budget.py <file.s> <mangled kernel name> [first_line last_line]

Categories: fma64 (v_fma_f64 / v_mul_f64 / v_add_f64: 4 issue cycles per wave on the 16-lane fp64 pipe; packed v_pk_*
counted in pk), mfma, dpp (v_*_dpp, v_permlane*), valu32 (every other VALU op), ds_read / ds_write, vmem (global_* /
buffer_* / flat_*), smem (s_load_*), salu, waitcnt, barrier, branch.  Dev tool (no GPU needed); the table under
profiles/r06/isa_budget.txt is made with it."""
import collections
import re
import sys


def category(op):
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_sleep'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_endpgm') or op.startswith('s_setpc'):
        return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_read') or op.startswith('ds_load') or op.startswith('ds_bpermute') or op.startswith('ds_swizzle'):
        return 'ds_read'
    if op.startswith('ds_'):
        return 'ds_write'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'mfma'
    if '_dpp' in op or op.startswith('v_permlane') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'):
        return 'dpp/lane'
    if op.startswith('v_pk_'):
        return 'pk64' if 'f64' in op or op in ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32') else 'valu32'
    if op.startswith('v_fma_f64') or op.startswith('v_mul_f64') or op.startswith('v_add_f64') or op.startswith('v_fmac_f64'):
        return 'fma64'
    if op.startswith('v_') and 'f64' in op:
        return 'other64'
    if op.startswith('v_accvgpr'):
        return 'acc mov'
    if op.startswith('v_'):
        return 'valu32'
    return 'other'


def blocks(lines):
    cur, out = ('entry', 0), []
    counts = collections.Counter()
    for i, raw in enumerate(lines):
        line = raw.split(';')[0].rstrip()
        m = re.match(r'^(\.LBB\d+_\d+):', line)
        if m:
            out.append((cur, counts))
            cur, counts = (m.group(1), i), collections.Counter()
            continue
        if re.match(r'^; %bb\.\d+', raw):
            out.append((cur, counts))
            cur, counts = (raw.split(':')[0].strip('; ').strip(), i), collections.Counter()
            continue
        s = line.strip()
        if not s or s.startswith('.') or s.endswith(':'):
            continue
        op = s.split()[0]
        counts[category(op)] += 1
    out.append((cur, counts))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    text = open(path).read().split('\n')
    start = next(i for i, l in enumerate(text) if l.startswith(name + ':'))
    end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
    body = text[start:end]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
    cats = ['fma64', 'pk64', 'other64', 'mfma', 'dpp/lane', 'valu32', 'acc mov', 'ds_read', 'ds_write', 'vmem', 'smem', 'salu',
            'wait', 'barrier', 'branch']
    print('%-14s %6s ' % ('block', 'line') + ' '.join('%8s' % c for c in cats))
    total = collections.Counter()
    for (label, line), counts in blocks(body):
        if not counts or not lo <= line < hi:
            continue
        total.update(counts)
        print('%-14s %6d ' % (label, line) + ' '.join('%8d' % counts.get(c, 0) for c in cats))
    print('%-14s %6s ' % ('sum', '') + ' '.join('%8d' % total.get(c, 0) for c in cats))


if __name__ == '__main__':
    main()
