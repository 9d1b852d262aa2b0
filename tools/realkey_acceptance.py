#!/usr/bin/env python3
"""One-command acceptance check against Vocdoni's REAL proving key -- the blob this build has never seen (`.MISSING_LARGE_BLOBS:1-3`; DESIGN.md section 7: "a shared
misreading of the snarkjs .zkey layout by setup, loader and oracle parser would pass every test here and fail on Vocdoni's key").  Whoever holds
`artifacts/zkCensus/dev/160/proving_key.zkey` runs

    python tools/realkey_acceptance.py /path/to/proving_key.zkey                     (needs an MI355X for steps 4-6; --parse-only stops after step 3)

and gets a PASS / FAIL line per step:

  1  sha256 of the file            == the published one (artifacts/zkCensus/dev/circuits-info.md:5, committed copy tests/golden/ref/circuits-info.md)
  2  section 2 / section 3         alpha1, beta2, gamma2, delta2, the pairing e(alpha1, beta2) against vk_alphabeta_12, and the nPublic + 1 IC points, read with THIS build's reading of the format (Montgomery, little endian, G2 as
                                   x.c0 x.c1 y.c0 y.c1), == the committed verification_key.json (verification_key.json:1-128) -- the first place a misreading would show
  3  shapes                        nVars 82754, nPublic 8, domain 2^17 (ZkFranchiseProofCircuit(160)); section sizes consistent with them; beta1 / delta1 on the curve
  4  groth16.fullProve             of the reference's own inputs_example.json through the key, on the GPU (native witness generator + HIP prover)
  5  public signals                == the reference's signals.json
  6  verifier                      the pinned verifier (zkc_verify: it accepts the reference's own proof.json) accepts the new proof UNDER THE REFERENCE'S verification key

Steps 1-3 are host-only and are exercised in the CPU suite against a key from the test-only setup and that key's own verification key
(tests/test_realkey_acceptance_cpu.py), so the comparison code itself is known to work before it meets the real blob.  --vkey / --inputs / --signals / --sha256 override the
committed reference fixtures (that is how the CPU test points the script at a test key)."""
import argparse
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'tests', 'golden', 'ref')
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
RINV_Q = pow(1 << 256, -1, Q)


def published_sha256(name='proving_key.zkey'):
    for line in open(os.path.join(REF, 'circuits-info.md')):
        parts = line.split()
        if len(parts) == 2 and parts[1].endswith(name):
            return parts[0]
    raise SystemExit('circuits-info.md has no line for ' + name)


def sections(buf):
    """snarkjs binfile: magic, u32 version, u32 nSections, then (u32 id, u64 size, payload)*  ->  {id: (offset, size)}"""
    if buf[:4] != b'zkey':
        raise ValueError('not a .zkey file (magic %r)' % buf[:4])
    ver, ns = struct.unpack_from('<II', buf, 4)
    out, p = {}, 12
    for _ in range(ns):
        sid, size = struct.unpack_from('<IQ', buf, p); p += 12
        if p + size > len(buf):
            raise ValueError('section %d runs past the end of the file' % sid)
        out[sid] = (p, size); p += size
    return ver, out


def fq(b):              # 32 bytes, Montgomery (R = 2^256), little endian -> standard integer
    return int.from_bytes(b, 'little') * RINV_Q % Q


def g1(buf, o):
    return [fq(buf[o:o + 32]), fq(buf[o + 32:o + 64])]


def g2(buf, o):
    return [[fq(buf[o:o + 32]), fq(buf[o + 32:o + 64])], [fq(buf[o + 64:o + 96]), fq(buf[o + 96:o + 128])]]


def on_curve_g1(p):
    return p == [0, 0] or (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0


def parse_key(buf):
    """-> dict(nVars, nPublic, domainSize, alpha1, beta1, beta2, gamma2, delta1, delta2, IC, section sizes); raises ValueError on a malformed file"""
    ver, sec = sections(buf)
    for need in range(1, 10):
        if need not in sec:
            raise ValueError('section %d is missing' % need)
    if struct.unpack_from('<I', buf, sec[1][0])[0] != 1:
        raise ValueError('not a Groth16 key (protocol id in section 1)')
    h = sec[2][0]
    n8q = struct.unpack_from('<I', buf, h)[0]; q = int.from_bytes(buf[h + 4:h + 4 + n8q], 'little')
    n8r = struct.unpack_from('<I', buf, h + 4 + n8q)[0]; r = int.from_bytes(buf[h + 8 + n8q:h + 8 + n8q + n8r], 'little')
    if (n8q, n8r, q, r) != (32, 32, Q, R):
        raise ValueError('not a BN254 key (field sizes / primes in section 2)')
    o = h + 8 + n8q + n8r
    nVars, nPub, dom = struct.unpack_from('<III', buf, o); o += 12
    k = {'version': ver, 'nVars': nVars, 'nPublic': nPub, 'domainSize': dom, 'sections': {i: s for i, (_, s) in sec.items()}}
    k['alpha1'] = g1(buf, o); k['beta1'] = g1(buf, o + 64); k['beta2'] = g2(buf, o + 128); k['gamma2'] = g2(buf, o + 256); k['delta1'] = g1(buf, o + 384); k['delta2'] = g2(buf, o + 448)
    k['IC'] = [g1(buf, sec[3][0] + 64 * i) for i in range(nPub + 1)]
    k['nCoeffs'] = struct.unpack_from('<I', buf, sec[4][0])[0]
    return k


def vk_points(vk):
    i = lambda p: [int(p[0]), int(p[1])]
    j = lambda p: [[int(p[0][0]), int(p[0][1])], [int(p[1][0]), int(p[1][1])]]
    return {'alpha1': i(vk['vk_alpha_1']), 'beta2': j(vk['vk_beta_2']), 'gamma2': j(vk['vk_gamma_2']), 'delta2': j(vk['vk_delta_2']), 'IC': [i(p) for p in vk['IC']], 'nPublic': int(vk['nPublic'])}


class Report:
    def __init__(self):
        self.rows = []

    def add(self, step, ok, what, detail=''):
        self.rows.append({'step': step, 'ok': bool(ok), 'what': what, 'detail': detail})
        print('%-4s step %s  %s%s' % ('PASS' if ok else 'FAIL', step, what, ('  -- ' + detail) if detail else ''), flush=True)
        return bool(ok)

    @property
    def ok(self):
        return all(r['ok'] for r in self.rows)


def check_file(buf, vk, want_sha, shape, rep):
    """steps 1-3 (host only)"""
    sha = hashlib.sha256(buf).hexdigest()
    rep.add(1, want_sha is None or sha == want_sha, 'sha256 of the key file' + ('' if want_sha else ' (no published value given: not compared)'), sha if want_sha in (None, sha) else '%s, published %s' % (sha, want_sha))
    try:
        k = parse_key(buf)
    except (ValueError, struct.error) as e:
        rep.add(2, False, 'the file parses as a snarkjs Groth16 .zkey', str(e))
        return None
    v = vk_points(vk)
    for name in ('alpha1', 'beta2', 'gamma2', 'delta2'):
        rep.add(2, k[name] == v[name], 'section 2 %s == verification key' % name, '' if k[name] == v[name] else 'key file %s..., verification key %s...' % (str(k[name])[:40], str(v[name])[:40]))
    if 'vk_alphabeta_12' in vk:         # e(alpha1, beta2) of the KEY FILE's points through the product's host pairing (no GPU) == the member snarkjs wrote into the verification key
        try:
            import ctypes
            sys.path.insert(0, ROOT)
            from zkcensus_amd import _native
            le = lambda x: int(x).to_bytes(32, 'little')
            out = ctypes.create_string_buffer(384)
            rc = _native.load().zkc_pairing_bin(le(k['alpha1'][0]) + le(k['alpha1'][1]), b''.join(le(c) for pair in k['beta2'] for c in pair), out)
            got = [str(int.from_bytes(out.raw[32 * i:32 * i + 32], 'little')) for i in range(12)]
            want12 = [str(x) for h in vk['vk_alphabeta_12'] for c in h for x in c]
            rep.add(2, rc == 0 and got == want12, 'e(alpha1, beta2) of the key file == vk_alphabeta_12 of the verification key')
        except OSError as e:
            rep.add(2, False, 'e(alpha1, beta2) of the key file == vk_alphabeta_12', 'libzkcensus.so not loadable: %s' % e)
    rep.add(2, k['nPublic'] == v['nPublic'] and k['IC'] == v['IC'], 'section 3: the %d IC points == verification key' % (k['nPublic'] + 1),
            '' if k['IC'] == v['IC'] else 'first mismatch at IC[%d]' % next((i for i, (a, b) in enumerate(zip(k['IC'], v['IC'])) if a != b), min(len(k['IC']), len(v['IC']))))
    nV, nP, dom = k['nVars'], k['nPublic'], k['domainSize']
    if shape:
        rep.add(3, (nV, nP, dom) == shape, 'shape: nVars %d, nPublic %d, domain %d' % shape, '' if (nV, nP, dom) == shape else 'the file says %d, %d, %d' % (nV, nP, dom))
    s = k['sections']
    want = {3: 64 * (nP + 1), 4: 4 + 44 * k['nCoeffs'], 5: 64 * nV, 6: 64 * nV, 7: 128 * nV, 8: 64 * (nV - nP - 1), 9: 64 * dom}
    bad = {i: (s[i], w) for i, w in want.items() if s[i] != w}
    rep.add(3, not bad and dom & (dom - 1) == 0, 'section sizes follow from (nVars, nPublic, domain, nCoeffs)', '' if not bad else 'section: (size, expected) ' + str(bad))
    rep.add(3, on_curve_g1(k['alpha1']) and on_curve_g1(k['beta1']) and on_curve_g1(k['delta1']) and all(on_curve_g1(p) for p in k['IC']), 'alpha1, beta1, delta1, IC lie on y^2 = x^3 + 3 after leaving Montgomery form')
    return k


def prove_and_verify(zkey_path, vk, inputs, signals, nLevels, rep):
    """steps 4-6 (GPU)"""
    sys.path.insert(0, ROOT)
    import zkcensus_amd                              # noqa: F401  (fails loudly without the HIP library / a GPU)
    from zkcensus_amd import groth16
    try:
        out = groth16.fullProve(inputs, None, zkey_path, nLevels=nLevels)
    except Exception as e:                          # noqa: BLE001 -- the report names the failure
        rep.add(4, False, 'groth16.fullProve(inputs_example.json) through the key', repr(e))
        return
    rep.add(4, True, 'groth16.fullProve(inputs_example.json) through the key')
    rep.add(5, out['publicSignals'] == [str(x) for x in signals], 'public signals == signals.json', '' if out['publicSignals'] == [str(x) for x in signals] else str(out['publicSignals']))
    ok = groth16.verify(vk, out['publicSignals'], out['proof'])
    rep.add(6, ok, "the pinned verifier accepts the proof under the reference's verification_key.json")
    bad = dict(out['proof']); bad['pi_a'] = [out['proof']['pi_a'][1], out['proof']['pi_a'][0], '1']
    try:
        rejected = not groth16.verify(vk, out['publicSignals'], bad)
    except Exception:                               # noqa: BLE001 -- a point off the curve is refused with an error: also a rejection
        rejected = True
    rep.add(6, rejected, 'and rejects the same proof with pi_a mangled (the verifier is not vacuous)')


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('zkey')
    ap.add_argument('--vkey', default=os.path.join(REF, 'verification_key.json'))
    ap.add_argument('--inputs', default=os.path.join(REF, 'inputs_example.json'))
    ap.add_argument('--signals', default=os.path.join(REF, 'signals.json'))
    ap.add_argument('--sha256', default=None, help="expected sha256 (default: the reference's published one; 'none' skips the comparison)")
    ap.add_argument('--nlevels', type=int, default=160)
    ap.add_argument('--any-shape', action='store_true', help='do not insist on 82754 / 8 / 2^17 (test keys of other depths)')
    ap.add_argument('--parse-only', action='store_true', help='steps 1-3 only (no GPU)')
    ap.add_argument('--json', default=None, help='write the report here')
    a = ap.parse_args(argv)
    buf = open(a.zkey, 'rb').read()
    vk = json.load(open(a.vkey))
    want = None if a.sha256 == 'none' else (a.sha256 or published_sha256())
    rep = Report()
    k = check_file(buf, vk, want, None if a.any_shape else (82754, 8, 1 << 17), rep)
    if k is not None and not a.parse_only:
        prove_and_verify(a.zkey, vk, json.load(open(a.inputs)), json.load(open(a.signals)), a.nlevels, rep)
    print('ACCEPTED' if rep.ok else 'NOT ACCEPTED')
    if a.json:
        json.dump({'accepted': rep.ok, 'steps': rep.rows}, open(a.json, 'w'), indent=1)
    return 0 if rep.ok else 1


if __name__ == '__main__':
    sys.exit(main())
