#!/usr/bin/env python3
"""Generate tests/golden/witness_vectors.json by running the REFERENCE's compiled witness calculator
(/root/reference/artifacts/zkCensus/dev/160/circuit.wasm, circom 2.1.5) under Node through tools/wasm_witness.js.
Runs in the build container only (the wasm does not travel).  For each voter the fixture keeps: the 12-key input
object (sibling lists stripped of trailing zeros), sha256 of the 82754 x 32-byte LE witness, the 8 public signals
and ~256 sampled (index, value) wires.  Negative vectors keep the wasm's error code and message."""
import json, os, random, subprocess, sys, hashlib, tempfile, copy
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth_voter import make_voter
import circuit_model as cm
WASM = '/root/reference/artifacts/zkCensus/dev/160/circuit.wasm'


def strip(v):
    v = list(v)
    while v and v[-1] == '0':
        v.pop()
    return v


def main():
    rng = random.Random(0x5A4B43454E535553)
    ex = json.load(open('/root/reference/artifacts/zkCensus/dev/160/inputs_example.json'))
    A = rng.getrandbits(160)
    voters = [
        ('example', ex),
        ('d7_12', make_voter(rng, depth_c=7, depth_s=12)),
        ('d160_3', make_voter(rng, depth_c=160, depth_s=3, zero_frac=0.5)),
        ('single_leaf', make_voter(rng, depth_c=0, depth_s=1)),
        ('d20_24', make_voter(rng, depth_c=20, depth_s=24, address=A)),
        ('d24_20_compl', make_voter(rng, depth_c=24, depth_s=20, address=A ^ ((1 << 160) - 1))),
        ('big_weight_eq', make_voter(rng, depth_c=3, depth_s=160, zero_frac=0.5, avail=(1 << 200) + 12345, vote=(1 << 200) + 12345)),
        ('weight_2p251', make_voter(rng, depth_c=1, depth_s=0, avail=(1 << 251) + 99, vote=5)),
        ('d159_158_dense', make_voter(rng, depth_c=159, depth_s=158, zero_frac=0.0, avail=1, vote=0)),
        ('addr_all_ones', make_voter(rng, depth_c=2, depth_s=2, zero_frac=0.0, address=(1 << 160) - 1)),
        ('addr_one', make_voter(rng, depth_c=33, depth_s=65, zero_frac=0.6, address=1)),
        ('field_addr', make_voter(rng, depth_c=9, depth_s=4, address=rng.randrange(cm.R))),
        ('addr_r_minus_1', make_voter(rng, depth_c=6, depth_s=11, address=cm.R - 1)),
        ('d160_160', make_voter(rng, depth_c=160, depth_s=160, zero_frac=0.5, address=rng.randrange(cm.R))),
        ('d158_158', make_voter(rng, depth_c=158, depth_s=158, zero_frac=0.5)),
        ('d157_159', make_voter(rng, depth_c=157, depth_s=159, zero_frac=0.9)),
        ('d13_17', make_voter(rng, depth_c=13, depth_s=17, zero_frac=0.0)),
        ('zero_weight_vote', make_voter(rng, depth_c=5, depth_s=5, avail=77, vote=0)),
    ]
    # [r3] a dozen more with the depths a real census has (2^10 .. 2^20 voters: leaves 10-20 levels down) and random weights, from their own generator so that the
    # eighteen above keep their values
    rng2 = random.Random(0x7A6B43454E535533)
    for i in range(12):
        avail = rng2.randrange(1, 1 << rng2.choice((8, 64, 200)))
        voters.append(('census_%02d' % i, make_voter(rng2, depth_c=rng2.randrange(10, 21), depth_s=rng2.randrange(10, 21), zero_frac=rng2.choice((0.0, 0.1, 0.5)), avail=avail, vote=rng2.randrange(0, avail + 1))))
    base = voters[1][1]

    def mut(**kw):
        v = copy.deepcopy(base)
        v.update(kw)
        return v
    bad_last = copy.deepcopy(base)
    bad_last['censusSiblings'][160] = '5'
    negs = [('weight_exceeds', mut(voteWeight=str(int(base['availableWeight']) + 1))),
            ('bad_census_root', mut(censusRoot=str((int(base['censusRoot']) + 1) % cm.R))),
            ('bad_sik_root', mut(sikRoot=str((int(base['sikRoot']) + 1) % cm.R))),
            ('bad_nullifier', mut(nullifier=str((int(base['nullifier']) + 1) % cm.R))),
            ('last_sibling_nonzero', bad_last)]
    # several violations in one voter: the wasm stops at the FIRST assert it reaches, and a drop-in reports that one.  Every pair and triple of the six
    # assert sites, and all six together.
    import itertools
    muts = {'weight': lambda v: v.update(voteWeight=str(int(v['availableWeight']) + 1)),
            'sik': lambda v: v.update(sikRoot=str((int(v['sikRoot']) + 1) % cm.R)),
            'census': lambda v: v.update(censusRoot=str((int(v['censusRoot']) + 1) % cm.R)),
            'null': lambda v: v.update(nullifier=str((int(v['nullifier']) + 1) % cm.R)),
            'lastc': lambda v: v['censusSiblings'].__setitem__(160, '5'),
            'lasts': lambda v: v['sikSiblings'].__setitem__(160, '7')}
    negs.append(('lasts', mut()))
    muts['lasts'](negs[-1][1])
    for r in (2, 3, 6):
        for combo in itertools.combinations(muts, r):
            v = copy.deepcopy(base)
            for k in combo:
                muts[k](v)
            negs.append(('+'.join(combo), v))
    allv = [v for _, v in voters] + [v for _, v in negs]
    with tempfile.TemporaryDirectory() as td:
        json.dump(allv, open(td + '/in.json', 'w'))
        st = json.loads(subprocess.check_output(['node', os.path.join(ROOT, 'tools/wasm_witness.js'), WASM, td + '/in.json', td + '/w']))
        out = {'nLevels': 160, 'nWires': 82754,
               'generator': 'tools/make_golden.py over the reference circuit.wasm (sha256 80a73567..., circuits-info.md:7)',
               'vectors': [], 'negative': []}
        for i, (name, v) in enumerate(voters):
            assert st[i]['ok'], (name, st[i])
            w = open('%s/w.%d' % (td, i), 'rb').read()
            assert hashlib.sha256(w).hexdigest() == st[i]['sha256']
            srng = random.Random(i)
            idx = sorted(set([0, 333, 334, 335, 336, 594, 838, 39303, 39304, 39549, 39793, 39794, 41138, 41139, 41140, 41391,
                              41687, 41948, 82753] + [srng.randrange(82754) for _ in range(237)]))
            vv = dict(v)
            vv['censusSiblings'] = strip(v['censusSiblings'])
            vv['sikSiblings'] = strip(v['sikSiblings'])
            out['vectors'].append({'name': name, 'inputs': vv, 'sha256': st[i]['sha256'],
                                   'public': [str(int.from_bytes(w[32 * k:32 * k + 32], 'little')) for k in range(1, 9)],
                                   'samples': [[k, str(int.from_bytes(w[32 * k:32 * k + 32], 'little'))] for k in idx]})
        for j, (name, v) in enumerate(negs):
            s = st[len(voters) + j]
            assert not s['ok'], name
            vv = dict(v)
            vv['censusSiblings'] = strip(v['censusSiblings'])
            vv['sikSiblings'] = strip(v['sikSiblings'])
            out['negative'].append({'name': name, 'inputs': vv, 'wasm_code': s['code'], 'wasm_msg': s['msg']})
    json.dump(out, open(os.path.join(ROOT, 'tests/golden/witness_vectors.json'), 'w'), indent=0)
    print('wrote', len(out['vectors']), 'vectors,', len(out['negative']), 'negative')
    for n in out['negative']:
        print(n['name'], n['wasm_code'], n['wasm_msg'].replace('\n', ' | '))


if __name__ == '__main__':
    main()
