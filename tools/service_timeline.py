#!/usr/bin/env python3
"""What the GPU did while the proving service was under load, from a rocprofv3 --kernel-trace CSV: per kernel name the launches, total and mean duration, and for the whole
trace the wall span, the time with NO kernel resident and the mean number of kernels resident at once.  usage: service_timeline.py <kernel_trace.csv> [skip-first-fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkc::', ''), r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows)
t_lo = ks[0][0] + (ks[-1][1] - ks[0][0]) * skip                 # steady state: the last part of the trace
ks = [k for k in ks if k[0] >= t_lo]
span = ks[-1][1] - ks[0][0]
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, st in ks:
    by[n][0] += 1; by[n][1] += e - s
ev = sorted([(s, 1) for s, e, *_ in ks] + [(e, -1) for s, e, *_ in ks])
cur = 0; last = ev[0][0]; idle = 0; area = 0
for t, d in ev:
    if cur == 0: idle += t - last
    area += cur * (t - last); cur += d; last = t
print('span %.2f ms, %d kernels, %d queues, %d streams; no kernel resident %.2f ms (%.1f %%); mean kernels resident %.2f' % (
    span / 1e6, len(ks), len(set(k[3] for k in ks)), len(set(k[4] for k in ks)), idle / 1e6, 100 * idle / span, area / span))
print('%-44s %7s %10s %9s %7s' % ('kernel', 'calls', 'total ms', 'mean us', '% span'))
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%-44s %7d %10.2f %9.1f %7.1f' % (n[:44], c, t / 1e6, t / c / 1e3, 100 * t / span))
