#!/usr/bin/env python3
"""Instruction histogram of one kernel's basic blocks from hipcc's device assembly (hipcc -S --cuda-device-only).

usage: isa_histogram.py file.s <substring of the mangled kernel name> [--min-block N]
Prints, for every basic block with at least N instructions, its label, instruction count and the split into v_mad_u64_u32 / other
VALU / SALU / memory / wait, and which label a trailing branch goes back to (loops).  Used for profiles/r02_accumulate_isa_histogram.txt:
the bound quoted for zkc_msm_accumulate29 in bench.py (roofline.valu) is the block mix of its mixed-addition loop priced with the rates of
tools/probe/rate_probe.hip."""
import collections, re, sys


def blocks_of(path, needle):
    name, cur, out, order = None, None, {}, []
    for line in open(path):
        s = line.strip()
        if name is None:
            if re.match(r'^[_A-Za-z0-9.$]+:', s) and needle in s and not s.startswith('.'):
                name = s.split(':')[0]; cur = 'entry'; out[cur] = []; order.append(cur)
            continue
        if s.startswith('.Lfunc_end') or s.startswith('.section') or s.startswith('.rodata'):
            break
        m = re.match(r'^(\.LBB[0-9_]+):', s)
        if m:
            cur = m.group(1); out[cur] = []; order.append(cur); continue
        if not s or s.startswith(';') or s.startswith('.') or s.startswith('//'):
            continue
        ins = s.split(';')[0].strip()
        if ins:
            out[cur].append(ins)
    return name, order, out


def classify(ins):
    op = ins.split()[0]
    if op == 'v_mad_u64_u32':
        return 'mad_u64_u32'
    if op.startswith('v_mul_lo_u32') or op.startswith('v_mul_hi_u32'):
        return 'mul32'
    if op.startswith('v_'):
        return 'valu_other'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    return 'other'


def main():
    path, needle = sys.argv[1], sys.argv[2]
    minb = int(sys.argv[sys.argv.index('--min-block') + 1]) if '--min-block' in sys.argv else 40
    name, order, blocks = blocks_of(path, needle)
    print('kernel:', name)
    pos = {b: i for i, b in enumerate(order)}
    tot = collections.Counter()
    for b in order:
        ins = blocks[b]
        c = collections.Counter(classify(i) for i in ins)
        tot.update(c)
        back = ''
        for i in ins[-3:]:
            m = re.search(r'(s_cbranch_\w+|s_branch)\s+(\.LBB[0-9_]+)', i)
            if m and m.group(2) in pos and pos[m.group(2)] <= pos[b]:
                back = ' -> back to %s (loop of %d blocks)' % (m.group(2), pos[b] - pos[m.group(2)] + 1)
        if len(ins) >= minb or back:
            print('%-12s %5d instr | mad_u64_u32 %4d mul32 %3d valu_other %4d salu %4d vmem %3d lds %3d wait %3d%s' % (
                b, len(ins), c['mad_u64_u32'], c['mul32'], c['valu_other'], c['salu'], c['vmem'], c['lds'], c['wait'], back))
            if '--ops' in sys.argv and len(ins) >= minb:
                ops = collections.Counter(i.split()[0] for i in ins)
                print('             ' + ', '.join('%s %d' % kv for kv in ops.most_common(14)))
    print('whole kernel: %d instr | ' % sum(tot.values()) + ' '.join('%s %d' % kv for kv in sorted(tot.items())))


if __name__ == '__main__':
    main()
