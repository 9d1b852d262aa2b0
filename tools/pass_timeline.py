#!/usr/bin/env python3
"""Timeline of ONE steady-state pipeline pass from a rocprofv3 --kernel-trace CSV: every kernel between the starts of two consecutive G1 accumulation launches in the
middle of a timed step, with its start (ms after the first accumulation's start), duration and stream, plus how much of the window has no VALU-heavy kernel running
(accumulations, transforms, reductions).  usage: pass_timeline.py <kernel_trace.csv> [which accumulate launch, default 12]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkc::', ''), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows if 'zkc' in r['Kernel_Name']]
ks.sort()
acc = [k for k in ks if k[2].startswith('zkc_msm_accumulate29<')]
t0, t1 = acc[which][0], acc[which + 1][0]
print('pass window %.3f ms (accumulate launch %d -> %d)' % ((t1 - t0) / 1e6, which, which + 1))
heavy = ('zkc_msm_accumulate29', 'zkc_ntt_', 'zkc_msm_window29', 'zkc_msm_merge29', 'zkc_msm_final29', 'zkc_matvec_jds', 'zkc_join_abc')
ev = []
for s, e, n, q in ks:
    if e < t0 or s > t1:
        continue
    print('%8.3f  +%7.3f ms  q%-3s %s' % ((s - t0) / 1e6, (e - s) / 1e6, q, n[:60]))
    if n.startswith(heavy):
        ev.append((max(s, t0), min(e, t1)))
ev.sort(); covered = 0; cur = None
for s, e in ev:
    if cur is None or s > cur[1]:
        if cur: covered += cur[1] - cur[0]
        cur = [s, e]
    else:
        cur[1] = max(cur[1], e)
if cur: covered += cur[1] - cur[0]
print('window with no VALU-heavy kernel resident: %.3f ms of %.3f' % ((t1 - t0 - covered) / 1e6, (t1 - t0) / 1e6))
