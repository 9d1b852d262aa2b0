#!/usr/bin/env python3
"""Timeline of ONE proof (config 2) from a rocprofv3 --kernel-trace CSV of tools/latency.py: every kernel of the last inputs -> proof call, start and end relative to
the call's first kernel, with its queue (stream).  Shows which chain is the critical path of a single proof.
usage: single_proof_trace.py <kernel_trace.csv>
  collected with: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/latency.py"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkc::', '')[:46], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
fin = [i for i, r in enumerate(rows) if r[2].startswith(('zkc_finalize', 'zkc_blind_tree_g1_out'))]      # the kernel that writes piA / piC: the last one of a proof
# the profiled call of latency.py comes after the 12 timing rounds (each round: host witness + prove, two device-resident forms); take the LAST one-call fullprove of the rounds:
# walk back from the last blinding kernel that is preceded by a witness kernel within 6 ms
pick = None
for i in reversed(fin):
    t_end = rows[i][1]
    wit = [r for r in rows if r[2].startswith('zkc_witness_chains') and 0 < t_end - r[0] < 6_000_000]
    if wit:
        pick = (wit[0][0], t_end); break
t0, t1 = pick
print('one inputs -> proof call: %.3f ms from the first witness kernel to the end of the blinding kernel' % ((t1 - t0) / 1e6))
for r in rows:
    if t0 - 200_000 <= r[0] <= t1 + 300_000 and r[2].startswith(('zkc', '__amd_rocclr')):
        print('%7.3f .. %7.3f  %6.3f ms  %-46s q%s' % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[2], r[3]))
