#!/usr/bin/env python3
"""The ADDITIVE per-kernel view of one pipeline pass (the wall-duration CSV of rocprofv3 --kernel-trace misranks kernels that idle beside
the accumulation on another stream).  Input: the counter_collection.csv of
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -- python3 bench.py --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --no-verify
(tools/gpu/call.sh ... pmc:sq:...).  Per kernel, summed over the launches of the one 64-proof pass:
    valu_busy_Mcycles_per_simd = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs   (the counter is in quad-cycles, summed over the chip)
    share                      = that / the sum over all zkc kernels     -- VALU work adds up across streams, wall durations do not
    gui_active_Mcycles         = GRBM_GUI_ACTIVE / 8 XCDs               (wall cycles while the kernel was resident: inflated by whatever ran beside it)
usage: kernel_busy_table.py <counter_collection.csv> <out.json> [commit]"""
import collections, csv, json, subprocess, sys

src, dst = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
for r in csv.DictReader(open(src)):
    k = r['Kernel_Name']
    if 'zkc' not in k:
        continue
    name = k.split('(')[0].replace('void ', '').replace('zkc::', '')
    acc[name][r['Counter_Name']] += float(r['Counter_Value']); launches[name].add(r['Dispatch_Id'])
rows = []
for k, v in acc.items():
    busy = v.get('SQ_ACTIVE_INST_VALU', 0.0) * 4 / 1024 / 1e6
    rows.append({'kernel': k, 'launches': len(launches[k]), 'valu_busy_Mcycles_per_simd': round(busy, 3), 'valu_instructions_G': round(v.get('SQ_INSTS_VALU', 0.0) / 1e9, 3),
                 'waves': int(v.get('SQ_WAVES', 0)), 'gui_active_Mcycles': round(v.get('GRBM_GUI_ACTIVE', 0.0) / 8 / 1e6, 3)})
SETUP = ('zkc_msm_shift_bases', 'zkc_fold_mul', 'zkc_fold_gsum', 'zkc_g2_table29', 'zkc_tw29', 'zkc_bitrev_copy', 'zkc_poseidon_batch_kernel', 'zkc_witness_tmpl', 'zkc_fb4_build')
for r in rows:
    r['phase'] = 'key load / first use (once per key)' if r['kernel'].startswith(SETUP) else 'pass'
tot = sum(r['valu_busy_Mcycles_per_simd'] for r in rows if r['phase'] == 'pass') or 1.0
for r in rows:
    r['share_of_pass'] = round(r['valu_busy_Mcycles_per_simd'] / tot, 4) if r['phase'] == 'pass' else None
rows.sort(key=lambda r: (r['phase'] != 'pass', -r['valu_busy_Mcycles_per_simd']))
json.dump({'command': 'rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -- python3 bench.py --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --no-verify',
           'commit': commit, 'units': __doc__.split('Per kernel')[1].split('usage')[0].strip(), 'pass_valu_busy_Mcycles_per_simd': round(tot, 2),
           'pass_valu_busy_ms_at_2.15_GHz': round(tot / 2.15, 2), 'kernels': rows}, open(dst, 'w'), indent=1)
for r in rows:
    if r['phase'] == 'pass' and r['share_of_pass'] >= 0.002:
        print('%-44s %3d launches  %8.3f Mcyc  %5.1f %%' % (r['kernel'][:44], r['launches'], r['valu_busy_Mcycles_per_simd'], 100 * r['share_of_pass']))
print('one 64-proof pass: %.2f M VALU-busy cycles per SIMD = %.2f ms at 2.15 GHz' % (tot, tot / 2.15))
