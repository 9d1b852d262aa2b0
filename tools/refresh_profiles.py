#!/usr/bin/env python3
"""Copies the outputs of an evidence run of tools/gpu/call.sh (gpurun_out/<run>/) into profiles/ under this round's names and rebuilds the summaries
(PMC HBM traffic, additive per-kernel VALU-busy table, VALU model).  usage: refresh_profiles.py gpurun_out/<run> [tag=r03] [msm.s]
The evidence run:  bash tools/gpu/call.sh <run> bench rate_probe fieldmul latency stress verify_bench prof pmc:fetch:FETCH_SIZE pmc:write:WRITE_SIZE \\
                        pmc:sq:SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,GRBM_GUI_ACTIVE,SQ_WAVES py:tools/service_bench.py,64,8 bench200"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = sys.argv[1]; tag = sys.argv[2] if len(sys.argv) > 2 else 'r03'; P = os.path.join(ROOT, 'profiles')
commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=ROOT).stdout.strip()


def cp(a, b):
    src = os.path.join(run, a)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, '%s_%s' % (tag, b))); return True
    print('missing', a); return False


cp('bench_default.json', 'bench_default.json'); cp('latency.json', 'single_proof_latency.json'); cp('stress.json', 'config5_stress.json'); cp('verify_bench.json', 'batch_verify.json')
cp('rate_probe.txt', 'rate_probe.txt'); cp('fieldmul_probe.txt', 'fieldmul_probe.txt'); cp('bench_200.json', 'bench_sustained_200_steps.json')
for f in glob.glob(os.path.join(run, 'py_tools_service_bench*.out')):
    shutil.copy(f, os.path.join(P, tag + '_service_bench.json'))
ks = glob.glob(os.path.join(run, 'prof', '**', '*_kernel_stats.csv'), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(P, tag + '_bench_b1024_kernel_stats.csv'))
if os.path.isdir(os.path.join(run, 'pmc_fetch')) and os.path.isdir(os.path.join(run, 'pmc_write')):
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(run, 'pmc_fetch'), os.path.join(run, 'pmc_write'), '64', os.path.join(P, tag + '_pmc_hbm_traffic.json')])
want = {'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'GRBM_GUI_ACTIVE', 'SQ_WAVES'}
for f in glob.glob(os.path.join(run, 'pmc_sq', '**', '*counter_collection.csv'), recursive=True):
    names = {ln.split(',')[15].strip('"') for ln in open(f) if ln.count(',') > 16}
    if want <= names and len(names & {'SQ_WAIT_INST_ANY'}) == 0:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'kernel_busy_table.py'), f, os.path.join(P, tag + '_kernel_valu_busy_table.json'), commit])
        break
if len(sys.argv) > 3 and os.path.exists(os.path.join(P, tag + '_rate_probe.txt')):
    hist = os.path.join(P, tag + '_accumulate_isa_histogram.txt')
    if not os.path.exists(hist):
        hist = os.path.join(P, 'r02_accumulate_isa_histogram.txt')
    hot = None
    for line in open(hist):
        if line.startswith('# hot path'):
            hot = ','.join(w for w in line.replace('(', ' ').split() if w.startswith('.LBB'))
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'valu_model.py'), sys.argv[3], 'zkc_msm_accumulate29ILi2', os.path.join(P, tag + '_rate_probe.txt'),
                           os.path.join(P, tag + '_valu_model.json'), '--hot', hot])
b = os.path.join(P, tag + '_bench_default.json')
if os.path.exists(b):
    j = json.loads([ln for ln in open(b) if ln.startswith('{')][0])
    print('bench', j['value'], 'roofline frac', j['roofline']['frac'], 'valu', j['roofline']['valu']['frac'], 'cpu', (j.get('cpu_baseline') or {}).get('value'), 'commit', commit)
