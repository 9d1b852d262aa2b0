#!/usr/bin/env python3
"""FETCH_SIZE calibration for 64-byte random gathers: usage pmc_calibrate.py <probe stdout json> <rocprofv3 --pmc FETCH_SIZE dir> <out.json>"""
import csv, glob, json, os, sys
known = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
f = glob.glob(os.path.join(sys.argv[2], '**', '*counter_collection.csv'), recursive=True)[0]
val = {}
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE':
        k = 'gather64' if 'gather64' in r['Kernel_Name'] else 'stream16' if 'stream16' in r['Kernel_Name'] else None
        if k:
            val[k] = val.get(k, 0.0) + float(r['Counter_Value'])
doc = {'command': 'rocprofv3 --pmc FETCH_SIZE --output-format csv -- tools/probe/gather_probe (built on the box)', 'known': known,
       'FETCH_SIZE_KiB': val,
       'bytes_per_reported_byte': {k: known[k + '_known_bytes'] / (v * 1024) for k, v in val.items()},
       'note': 'stream16 reproduces the guide\'s gfx950 figure for wide coalesced streams (the counter reports half the bytes); gather64 is the factor for the access '
               'pattern of zkc_msm_accumulate29 (random 64-byte rows, four 16-byte loads per lane): tools/pmc_summary.py multiplies FETCH_SIZE by it'}
json.dump(doc, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(doc['bytes_per_reported_byte']))
