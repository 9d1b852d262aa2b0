#!/usr/bin/env python3
"""VALU issue model of the G1 bucket accumulation (zkc_msm_accumulate29), reproducible from two committed files:

  profiles/r02_rate_probe.txt               lane-operations/s per instruction on this part (tools/probe/rate_probe.hip, W = 3 waves per SIMD)
  profiles/r02_accumulate_isa_histogram.txt instruction mix per basic block of the kernel (tools/isa_histogram.py over hipcc -S output)

usage: valu_model.py <kernel.s> <kernel name needle> <rate_probe.txt> <out.json> --hot .LBBa,.LBBb,...
The hot path = the basic blocks one mixed addition executes (loop head, next-point gather + sign handling, the equal-x test and the ten products);
the cold blocks of the loop (doubling through generic code, first point of a segment, P + (-P)) are listed but not priced.
capacity = 1 / sum_i (count_i / rate_i) mixed additions per second; bench.py divides the measured rate by it (roofline.valu.frac).
Instructions the probe did not measure are priced at the FAST class rate, which can only lower the reported fraction."""
import collections, json, re, sys
sys.path.insert(0, __import__('os').path.dirname(__file__))
from isa_histogram import blocks_of

FAST = {'v_add_u32', 'v_and_b32', 'v_mov_b32'}                                   # measured ~50 T lane-ops/s at W = 3
SLOW = {'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u32_u24', 'v_lshl_add_u64', 'v_lshrrev_b64', 'v_alignbit_b32', 'v_add3_u32', 'v_lshl_add_u32', 'v_perm_b32',
        'v_add_co_u32', 'v_addc_co_u32', 'v_fma_f64'}                            # measured ~32-36 T lane-ops/s


def main():
    asm, needle, probe, out = sys.argv[1:5]
    hot = sys.argv[sys.argv.index('--hot') + 1].split(',')
    rates = {}
    for line in open(probe):
        m = re.match(r'^(\S.*?)\s+W=3\s+[\d.]+ ms\s+([\d.]+) T lane-ops/s', line)
        if m:
            rates[m.group(1).strip()] = float(m.group(2)) * 1e12
    r_mad = rates['v_mad_u64_u32']
    r_fast = sum(rates[k] for k in ('v_add_u32', 'v_and_b32')) / 2
    slow_meas = [rates[k] for k in ('v_mul_lo_u32', 'v_lshl_add_u64', 'v_lshrrev_b64', 'v_alignbit_b32', 'v_add3_u32') if k in rates]
    r_slow = sum(slow_meas) / len(slow_meas)
    name, order, blocks = blocks_of(asm, needle)
    cnt = collections.Counter(); unknown = collections.Counter(); per_block = {}
    for b in hot:
        c = collections.Counter()
        for ins in blocks[b]:
            op = ins.split()[0]; base = re.sub(r'_e(32|64)$', '', op)
            if base == 'v_mad_u64_u32': k = 'mad_u64_u32'
            elif base in SLOW: k = 'half_rate'
            elif base in FAST: k = 'full_rate'
            elif op.startswith('v_'): k = 'full_rate'; unknown[base] += 1           # unmeasured: priced fast (conservative for the fraction)
            else: k = 'scalar_or_memory'
            c[k] += 1
        per_block[b] = dict(c); cnt.update(c)
    valu = cnt['mad_u64_u32'] + cnt['half_rate'] + cnt['full_rate']
    t = cnt['mad_u64_u32'] / r_mad + cnt['half_rate'] / r_slow + cnt['full_rate'] / r_fast
    # the direct measurement: a dependency-free loop with the kernel's own mix (10 mad : 3 half-rate : 3 full-rate per 16 instructions) at every occupancy
    mix = {}
    for line in open(probe):
        m = re.match(r'^accumulation mix.*?W=(\d+)\s+[\d.]+ ms\s+([\d.]+) T lane-ops/s', line)
        if m:
            mix[int(m.group(1))] = float(m.group(2)) * 1e12
    mix_best = max(mix.values()) if mix else None
    doc = {'kernel': name, 'hot_blocks': hot, 'per_block': per_block,
           'instr_per_madd': valu, 'mad_u64_u32_per_madd': cnt['mad_u64_u32'], 'half_rate_per_madd': cnt['half_rate'], 'full_rate_per_madd': cnt['full_rate'],
           'scalar_or_memory_per_madd': cnt['scalar_or_memory'],
           'rate_mad_u64_u32': r_mad, 'rate_half_rate_class': r_slow, 'rate_valu32': r_fast, 'rates_measured_at': 'W = 3 waves per SIMD, all 1024 SIMDs busy (sclk 2.1-2.2 GHz under this load: profiles/r02_power_clock_trace.json)',
           'capacity_madd_per_s_from_single_op_rates': 1.0 / t,
           'mix_probe_lane_ops_per_s_by_waves_per_simd': mix, 'mix_probe_best': mix_best,
           'capacity_madd_per_s': (mix_best / valu) if mix_best else 1.0 / t,
           'capacity_note': 'capacity = best rate the mix probe sustains at any occupancy / VALU instructions of one mixed addition; the single-op rates give a lower '
                            'figure because the mix interleaves 2-cycle and 4-cycle instructions (one 4-cycle wave-instruction per SIMD at 2.15 GHz is 35 T lane-ops/s)',
           'unmeasured_ops_priced_fast': dict(unknown),
           'source': 'profiles/r02_rate_probe.txt + profiles/r02_accumulate_isa_histogram.txt via tools/valu_model.py'}
    json.dump(doc, open(out, 'w'), indent=1)
    print(json.dumps({k: doc[k] for k in ('instr_per_madd', 'mad_u64_u32_per_madd', 'half_rate_per_madd', 'full_rate_per_madd', 'capacity_madd_per_s', 'capacity_madd_per_s_from_single_op_rates', 'mix_probe_best')}))
    print('unmeasured (priced fast):', dict(unknown))


if __name__ == '__main__':
    main()
