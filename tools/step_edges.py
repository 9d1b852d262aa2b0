#!/usr/bin/env python3
"""Where a bench step's non-overlapped time goes: from a rocprofv3 --kernel-trace CSV, the kernels of the LAST step's first and last milliseconds.
usage: step_edges.py <kernel_trace.csv> [ms at each edge]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkc::', '')[:44], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
edge = float(sys.argv[2]) if len(sys.argv) > 2 else 18.0
acc = [r for r in rows if r[2].startswith('zkc_msm_accumulate29<')]
# steps are separated by gaps between G1 accumulations longer than 3x the median period
starts = [a[0] for a in acc]; per = sorted(b - a for a, b in zip(starts, starts[1:]))[len(starts) // 2]
cut = [i + 1 for i, (a, b) in enumerate(zip(starts, starts[1:])) if False]
# simpler: the last step = the last N accumulations where N = passes per step (count / steps is not known): find the last witness launch burst instead
# a step's first pass takes the wave-per-path witness kernel (one launch per step at batch 1024): the last such launch starts the last step
wit = [r for r in rows if r[2].startswith('zkc_witness_chains_wave')]
first_wit_last_step = wit[-1] if wit else None
t0 = first_wit_last_step[0] if first_wit_last_step else rows[0][0]
step = [r for r in rows if r[0] >= t0 - 1_000_000]
tend = max(r[1] for r in step)
print('last step: %.2f ms from its first witness kernel to its last kernel end; median G1-accumulation period %.2f ms' % ((tend - t0) / 1e6, per / 1e6))
a = [r for r in step if r[2].startswith('zkc_msm_accumulate29<')]
print('first G1 accumulation starts at +%.2f ms; last one ends %.2f ms before the end; %d accumulations, sum %.2f ms' % ((a[0][0] - t0) / 1e6, (tend - a[-1][1]) / 1e6, len(a), sum(x[1] - x[0] for x in a) / 1e6))
print('--- first %.0f ms ---' % edge)
for r in step:
    if r[0] - t0 < edge * 1e6 and r[1] - r[0] > 30_000:
        print('%8.2f .. %8.2f  %-44s q%s' % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, r[2], r[3]))
print('--- last %.0f ms ---' % (edge / 2))
for r in step:
    if tend - r[1] < edge / 2 * 1e6 and r[1] - r[0] > 3_000:
        print('%8.2f .. %8.2f  %-44s q%s' % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, r[2], r[3]))
