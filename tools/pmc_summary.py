"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_hbm_traffic.json.

usage: python tools/pmc_summary.py <dir with *_counter_collection.csv of the FETCH pass> <dir of the WRITE pass> <proofs per launch> <out.json>
The roofline kernel is the G1 bucket accumulation (zkc_msm_accumulate29); bench.py reads hbm_bytes_per_launch_uncorrected from the file.
"""
import csv, glob, json, os, sys, collections


def load(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0]); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']; acc[k][1] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    return {k: {'launches': len(seen[k]), 'counter_sum': v[1], 'per_launch': v[1] / max(1, len(seen[k]))} for k, v in acc.items() if k.startswith('zkc') or 'zkc::' in k}


def main():
    fdir, wdir, ppl, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = load(fdir), load(wdir)
    kern = [k for k in fe if 'zkc_msm_accumulate29<' in k][0]
    calib = {'gather64': 1.0}
    try:
        calib = json.load(open(os.path.join(os.path.dirname(os.path.abspath(out)), 'r02_pmc_fetch_calibration.json')))['bytes_per_reported_byte']
    except Exception:
        pass
    doc = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --batch %d --steps 1 --warmup 0 --no-cpu-baseline' % ppl,
           'kernel': kern, 'proofs_per_launch': ppl,
           'FETCH_SIZE_per_launch': fe[kern]['per_launch'], 'WRITE_SIZE_per_launch': wr[kern]['per_launch'],
           'hbm_bytes_per_launch_uncorrected': (fe[kern]['per_launch'] + wr[kern]['per_launch']) * 1024,
           'fetch_calibration_factor': calib.get('gather64', 1.0),
           'hbm_bytes_per_launch': (fe[kern]['per_launch'] * calib.get('gather64', 1.0) + wr[kern]['per_launch']) * 1024,
           'hbm_bytes_per_launch_with_stream_factor_2': (fe[kern]['per_launch'] * 2.0 + wr[kern]['per_launch']) * 1024,      # the guide's factor for WIDE COALESCED streams, for comparison only
           'note': 'counters are in KiB (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024).  The guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide '
                   'coalesced stream on gfx950, "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  That calibration: '
                   '(reproduced: profiles/r02_pmc_fetch_calibration.json, stream16 = 2.00); for this kernel\'s pattern -- random 64-byte rows, four 16-byte loads per lane -- '
                   'the same calibration run gives 0.95 bytes per reported byte (gather64), which is the factor applied to FETCH_SIZE here.',
           'all_zkc_kernels': {'pmc_fetch': fe, 'pmc_write': wr}}
    json.dump(doc, open(out, 'w'), indent=1)
    print(kern[:60], 'FETCH %.0f KiB WRITE %.0f KiB per launch -> %.3f GB' % (fe[kern]['per_launch'], wr[kern]['per_launch'], doc['hbm_bytes_per_launch_uncorrected'] / 1e9))


if __name__ == '__main__':
    main()
