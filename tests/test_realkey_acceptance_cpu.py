"""CPU: the host half (steps 1-3) of tools/realkey_acceptance.py -- the one-command check for whoever holds Vocdoni's real proving_key.zkey
(artifacts/zkCensus/dev/circuits-info.md:5; verification_key.json:1-128) -- exercised against keys from the test-only setup and THEIR verification keys, so that the
comparison code is known to accept what it should and to notice every kind of disagreement before it ever meets the real blob."""
import json, os, sys
import pytest
import oracle_lib as ol

sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
import realkey_acceptance as ra


def _run(args, capsys):
    rc = ra.main(args)
    return rc, capsys.readouterr().out


def test_published_hash_is_read_from_the_committed_fixture():
    assert ra.published_sha256() == 'e359b256e5e3c78acaccf8dab5dc4bea99a2f07b2a05e935b5ca658c714dea4a'
    assert ra.published_sha256('verification_key.json') == '235e55571812f8e324e73e37e53829db0c4ac8f68469b9b953876127c97b425f'
    import hashlib
    assert hashlib.sha256(open(ol.golden('ref/verification_key.json'), 'rb').read()).hexdigest() == ra.published_sha256('verification_key.json')


def test_parser_half_accepts_a_key_with_its_own_verification_key(tmp_path, capsys):
    from zkcensus_amd import setup
    _, z160, v160 = setup.ensure_test_artifacts(160)
    rc, out = _run([z160, '--vkey', v160, '--sha256', 'none', '--parse-only', '--json', str(tmp_path / 'rep.json')], capsys)      # default shape: 82754 / 8 / 2^17
    assert rc == 0 and 'ACCEPTED' in out and 'FAIL' not in out, out
    rep = json.load(open(tmp_path / 'rep.json'))
    assert rep['accepted'] and {r['step'] for r in rep['steps']} == {1, 2, 3}
    # the same key against the REFERENCE's verification key: alpha, beta, gamma, delta and IC all differ (another ceremony), the shape still matches; the published hash does not
    rc, out = _run([z160, '--parse-only'], capsys)
    assert rc == 1 and 'NOT ACCEPTED' in out
    assert 'FAIL step 1' in out and 'FAIL step 2  section 2 alpha1' in out and 'FAIL step 2  section 3' in out and 'PASS step 3  shape' in out, out
    assert 'FAIL step 2  e(alpha1, beta2) of the key file == vk_alphabeta_12' in out
    # what the points are compared with is the oracle's independent reading of the same file (tests only): both readers agree on the test key
    k = ra.parse_key(open(z160, 'rb').read())
    vko = ol.zkey_vk(open(z160, 'rb').read())
    assert ra.vk_points(json.load(open(v160)))['IC'] == k['IC'] and len(k['IC']) == 9
    le = lambda x: x.to_bytes(32, 'little'); g1 = lambda p: le(p[0]) + le(p[1]); g2 = lambda p: le(p[0][0]) + le(p[0][1]) + le(p[1][0]) + le(p[1][1])
    assert g1(k['alpha1']) + g2(k['beta2']) + g2(k['gamma2']) + g2(k['delta2']) + b''.join(g1(p) for p in k['IC']) == vko      # alpha1 | beta2 | gamma2 | delta2 | IC, standard form


def test_parser_half_notices_disagreements(tmp_path, capsys):
    from zkcensus_amd import setup
    _, z10, v10 = setup.ensure_test_artifacts(10)
    rc, out = _run([z10, '--vkey', v10, '--sha256', 'none', '--parse-only', '--any-shape'], capsys)
    assert rc == 0, out
    rc, out = _run([z10, '--vkey', v10, '--sha256', 'none', '--parse-only'], capsys)                  # an nLevels-10 key is not the 82754-wire circuit
    assert rc == 1 and 'FAIL step 3  shape' in out
    raw = bytearray(open(z10, 'rb').read())
    _, sec = ra.sections(bytes(raw))
    # one bit in the last IC point
    bad = bytearray(raw); bad[sec[3][0] + 64 * 8 + 5] ^= 1
    p = tmp_path / 'ic.zkey'; p.write_bytes(bad)
    rc, out = _run([str(p), '--vkey', v10, '--sha256', 'none', '--parse-only', '--any-shape'], capsys)
    assert rc == 1 and 'first mismatch at IC[8]' in out and 'FAIL step 3  alpha1, beta1, delta1, IC lie on' in out
    # gamma2 and delta2 swapped in the header (a reader that confused them would pass its own tests)
    h = sec[2][0] + 8 + 64 + 12
    bad = bytearray(raw); bad[h + 256:h + 384], bad[h + 448:h + 576] = raw[h + 448:h + 576], raw[h + 256:h + 384]
    p = tmp_path / 'gd.zkey'; p.write_bytes(bad)
    rc, out = _run([str(p), '--vkey', v10, '--sha256', 'none', '--parse-only', '--any-shape'], capsys)
    assert rc == 1 and 'FAIL step 2  section 2 gamma2' in out and 'FAIL step 2  section 2 delta2' in out and 'PASS step 2  section 2 alpha1' in out
    assert 'PASS step 2  e(alpha1, beta2) of the key file == vk_alphabeta_12' in out             # alpha and beta are untouched in that file
    # a truncated file and a file that is not a key
    p = tmp_path / 'short.zkey'; p.write_bytes(raw[:len(raw) // 2])
    rc, out = _run([str(p), '--vkey', v10, '--sha256', 'none', '--parse-only', '--any-shape'], capsys)
    assert rc == 1 and 'runs past the end of the file' in out
    p = tmp_path / 'no.zkey'; p.write_bytes(b'wtns' + bytes(100))
    rc, out = _run([str(p), '--vkey', v10, '--sha256', 'none', '--parse-only', '--any-shape'], capsys)
    assert rc == 1 and 'not a .zkey file' in out
    # a wrong expected hash
    rc, out = _run([z10, '--vkey', v10, '--sha256', '00' * 32, '--parse-only', '--any-shape'], capsys)
    assert rc == 1 and 'FAIL step 1' in out
