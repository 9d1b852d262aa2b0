"""Pins the CPU oracle (oracle/) against everything the reference holds for this path:
the reference wasm's witnesses (tests/golden/witness_vectors.json, made by tools/make_golden.py), the committed
public signals, the valid (proof, signals, verification_key) triple, Poseidon known answers and the
input-encoding known answers.  CPU only."""
import hashlib, json, copy
import pytest
import oracle_lib as ol

VEC = ol.load_json('witness_vectors.json')


def test_poseidon_kats():
    # BASELINE.md section 5 / SURVEY.md A.3-A.4
    assert ol.poseidon([1, 2]) == 7853200120776062878684798364095072458815029376092732009249414926327459813530
    assert ol.poseidon([1, 2, 3, 4]) == 18821383157269793795438455681495246036402687001665670618754263018637548127333
    assert ol.poseidon([0, 0]) == 14744269619966411208579211824598458697587494354926760081771325075741142829156
    assert ol.poseidon([0, 0, 1]) == 3108394280857290448796042949317662357879960495408018998613518544538624657019
    ex = ol.load_json('ref/inputs_example.json')
    sik = ol.poseidon([int(ex['address']), int(ex['password']), int(ex['signature'])])
    assert sik == 2763692874536778083565625622297672041243620578525557176716292711186292779392
    assert ol.poseidon([int(ex['signature']), int(ex['password']), int(ex['electionId'][0]), int(ex['electionId'][1])]) == int(ex['nullifier'])
    assert ol.poseidon([int(ex['address']), 10, 1]) == 16238148492350107382842142274532928432385429886098373207269294250724638770275


def test_example_witness_sha_and_public_signals():
    ex = ol.load_json('ref/inputs_example.json')
    rc, w = ol.witness(ex)
    assert rc == 0 and len(w) == 82754 * 32
    assert hashlib.sha256(w).hexdigest() == 'ebf5467e953a0427fa50c9a0b0521ac1c5c70684ef3603c75807c1bc4315e71b'
    sig = ol.load_json('ref/signals.json')
    assert [str(int.from_bytes(w[32 * i:32 * i + 32], 'little')) for i in range(1, 9)] == sig


@pytest.mark.parametrize('vec', VEC['vectors'], ids=[v['name'] for v in VEC['vectors']])
def test_witness_matches_reference_wasm(vec):
    rc, w = ol.witness(vec['inputs'])
    assert rc == 0
    assert hashlib.sha256(w).hexdigest() == vec['sha256']
    for k, v in vec['samples']:
        assert int.from_bytes(w[32 * k:32 * k + 32], 'little') == int(v)
    assert [str(int.from_bytes(w[32 * i:32 * i + 32], 'little')) for i in range(1, 9)] == vec['public']


EXPECT = {'weight_exceeds': 1, 'bad_sik_root': 2, 'bad_census_root': 3, 'bad_nullifier': 4, 'last_sibling_nonzero': 5, 'lasts': 7}


@pytest.mark.parametrize('vec', VEC['negative'], ids=[v['name'] for v in VEC['negative']])
def test_witness_rejects_like_reference_wasm(vec):
    """42 rejected voters: each assert alone, then every pair and triple of the six assert sites and all six at once.  The wasm stops at the first assert it
    reaches; the oracle's status must name that one (not merely "some assert failed")."""
    assert vec['wasm_code'] == 4          # the reference raises "assert failed" for every one
    rc, _ = ol.witness(vec['inputs'])
    assert rc == ol.status_of_wasm_message(vec['wasm_msg'])
    if vec['name'] in EXPECT:
        assert rc == EXPECT[vec['name']]


def test_status_text_is_the_reference_message():
    """zkc_witness_status_text (host only, no GPU) returns snarkjs's Error.message for the status: "Assert Failed.\\n" + the wasm's lines, byte for byte at nLevels 160"""
    from zkcensus_amd import _native
    lib = _native.load()
    seen = set()
    for vec in VEC['negative']:
        st = ol.status_of_wasm_message(vec['wasm_msg']); seen.add(st)
        assert lib.zkc_witness_status_text(160, st).decode() == 'Assert Failed.\n' + vec['wasm_msg'] + '\n'
    assert seen == {1, 2, 3, 4, 5, 7}
    import re
    for st in seen:                                                     # other depths: the same frames without the instance numbers of the dev/160 build
        assert lib.zkc_witness_status_text(10, st).decode() == re.sub(r'_\d+ line', ' line', lib.zkc_witness_status_text(160, st).decode())
    assert lib.zkc_witness_status_text(160, 0) is None and lib.zkc_witness_status_text(160, 8) is None and lib.zkc_witness_status_text(160, -1) is None
    assert b'field order' in lib.zkc_witness_status_text(160, 6)


def test_verifier_accepts_reference_triple_and_rejects_bitflips():
    vk = ol.load_json('ref/verification_key.json'); pr = ol.load_json('ref/proof.json'); sig = ol.load_json('ref/signals.json')
    assert vk['nPublic'] == 8 and len(vk['IC']) == 9
    assert ol.verify(vk, sig, pr)
    pb = bytearray(ol.proof_bytes(pr))
    for off in (0, 40, 64, 130, 200, 255):
        bad = bytearray(pb); bad[off] ^= 1
        assert not ol.verify(vk, sig, bytes(bad))
    for i in (0, 2, 7):
        s2 = list(sig); s2[i] = str((int(s2[i]) + 1) % ol.R)
        assert not ol.verify(vk, s2, pr)


def test_input_encoding_kats():
    """raw client values (ts_inputs/src/example.ts:340-346) -> inputs_example.json decimals
    (internal/helpers.go:17-34, internal/inputs.go:81-92, ts_inputs/src/arbo_utils.ts:10-33, ff.ts:3-18)."""
    ex = ol.load_json('ref/inputs_example.json')
    def arbo_hash(b):
        h = hashlib.sha256(b).digest()
        return [str(int.from_bytes(h[:16], 'little')), str(int.from_bytes(h[16:], 'little'))]
    assert arbo_hash(bytes.fromhex('7faeab7a7d250527d614e952ae8e446825bd1124c6def410844c7c383d1519a6')) == ex['electionId']
    assert arbo_hash(bytes([10])) == ex['voteHash']
    assert str(int.from_bytes(bytes.fromhex('032234DBb3B6dA8c11DDdc26338867C769e66B00'), 'little')) == ex['address']
    assert str(int.from_bytes(b'password123', 'big') % ol.R) == ex['password']
    sig = bytes.fromhex('7b6cac3c3b64d0b7fc10f0f6d4b8baf2548f246a748d25d8825becc1e2fa3c6e0a2654b042be487f0a352bc2c0577cde1440373197b4d93e09fc7502b61e9632')
    assert str(int.from_bytes(sig, 'big') % ol.R) == ex['signature']
    assert len(ex['censusSiblings']) == 161 and ex['censusSiblings'][-1] == '0'


def test_fixture_hashes():
    info = open(ol.golden('ref/circuits-info.md')).read()
    vk_sha = hashlib.sha256(open(ol.golden('ref/verification_key.json'), 'rb').read()).hexdigest()
    assert vk_sha in info
