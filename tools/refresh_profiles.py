#!/usr/bin/env python3
"""Copies the outputs of tools/gpu/r02_call_c.sh (gpurun_out/<run>/) into profiles/ under their round-2 names and rebuilds the summaries
(PMC HBM traffic, SQ counters, VALU model).  usage: refresh_profiles.py gpurun_out/<run> [msm.s]"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = sys.argv[1]; P = os.path.join(ROOT, 'profiles')
cp = lambda a, b: shutil.copy(os.path.join(run, a), os.path.join(P, b))
cp('bench_default.json', 'r02_bench_default.json'); cp('latency.json', 'r02_single_proof_latency.json'); cp('stress.json', 'r02_config5_stress.json')
cp('verify_bench.json', 'r02_batch_verify.json'); cp('rate_probe.txt', 'r02_rate_probe.txt')
shutil.copy(glob.glob(os.path.join(run, 'prof', '**', '*_kernel_stats.csv'), recursive=True)[0], os.path.join(P, 'r02_bench_b1024_kernel_stats.csv'))
subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(run, 'pmc_fetch'), os.path.join(run, 'pmc_write'), '96', os.path.join(P, 'r02_pmc_hbm_traffic.json')])
f = glob.glob(os.path.join(run, 'pmc_sq', '**', '*counter_collection.csv'), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']; acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
out = {}
for k, v in acc.items():
    if 'zkc' in k:
        d = {a: b / len(n[k]) for a, b in v.items()}; d['launches'] = len(n[k])
        if d.get('GRBM_GUI_ACTIVE'):
            d['valu_busy_frac_est'] = round((d['SQ_ACTIVE_INST_VALU'] * 4 / 1024) / (d['GRBM_GUI_ACTIVE'] / 8), 3)
        out[k.split('(')[0][-60:]] = d
json.dump({'command': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -- python3 bench.py --batch 96 --steps 1 --warmup 0 --no-cpu-baseline --no-verify',
           'units': 'per launch averages; SQ_* summed over the chip (SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES in quad-cycles per the microarch guide), GRBM_GUI_ACTIVE summed over the 8 XCDs',
           'valu_busy_frac_est': "(SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8): share of the kernel's cycles in which a SIMD's VALU is executing",
           'kernels': out}, open(os.path.join(P, 'r02_pmc_sq_accumulate.json'), 'w'), indent=1)
for k, d in out.items():
    if 'accumulate' in k or 'ntt' in k:
        print(k, d.get('valu_busy_frac_est'))
if len(sys.argv) > 2:
    hot = None
    for line in open(os.path.join(P, 'r02_accumulate_isa_histogram.txt')):
        if line.startswith('# hot path'):
            hot = ','.join(w for w in line.replace('(', ' ').split() if w.startswith('.LBB'))
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'valu_model.py'), sys.argv[2], 'zkc_msm_accumulate29ILi2', os.path.join(P, 'r02_rate_probe.txt'),
                           os.path.join(P, 'r02_valu_model.json'), '--hot', hot])
j = json.load(open(os.path.join(P, 'r02_bench_default.json')))
print('bench', j['value'], 'roofline frac', j['roofline']['frac'], 'achieved GB/s', j['roofline']['achieved'], 'valu', j['roofline']['valu']['achieved'], j['roofline']['valu']['frac'], 'cpu', j['cpu_baseline']['value'])
