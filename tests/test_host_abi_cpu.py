"""Host-only entry points of the C ABI that need no GPU: the snarkjs-compatible pairing value (pinned to the reference's own
verification_key.json), circuit selection by wasm hash (circuits-info.md:7), verifier strictness, blinding-scalar sampling, and the
rapidsnark-shaped entry point's argument checks that come before any GPU work."""
import ctypes, hashlib, json, os
import pytest
import oracle_lib as ol
import zkcensus_amd
from zkcensus_amd import _native, groth16


def test_pairing_matches_reference_vk_alphabeta_12():
    """verification_key.json of the reference carries e(alpha1, beta2) as snarkjs computed it (vk_alphabeta_12, :52): a known-answer
    test for the product's optimal-ate pairing + libff final exponentiation, and the value zkc_setup_from_r1cs now writes."""
    lib = _native.load()
    vk = ol.load_json('ref/verification_key.json')
    out = ctypes.create_string_buffer(384)
    assert lib.zkc_pairing_bin(ol.g1_json(vk['vk_alpha_1']), ol.g2_json(vk['vk_beta_2']), out) == 0
    got = [str(int.from_bytes(out.raw[32 * i:32 * i + 32], 'little')) for i in range(12)]
    assert got == [x for h in vk['vk_alphabeta_12'] for c in h for x in c]
    # bilinearity through the same function: e(2 alpha, beta) == e(alpha, beta)^2 is not expressible here without GT arithmetic, but
    # e(alpha, beta) must differ from e(IC0, beta), and a point off the curve is refused
    out2 = ctypes.create_string_buffer(384)
    assert lib.zkc_pairing_bin(ol.g1_json(vk['IC'][0]), ol.g2_json(vk['vk_beta_2']), out2) == 0 and out2.raw != out.raw
    bad = bytearray(ol.g1_json(vk['vk_alpha_1'])); bad[0] ^= 1
    assert lib.zkc_pairing_bin(bytes(bad), ol.g2_json(vk['vk_beta_2']), out2) != 0


def test_generated_verification_key_has_reference_schema(tmp_path):
    from zkcensus_amd import setup
    _, _, vp = setup.ensure_test_artifacts(10, directory=str(tmp_path))
    mine = json.load(open(vp)); ref = ol.load_json('ref/verification_key.json')
    assert list(mine.keys()) == list(ref.keys())
    ab = mine['vk_alphabeta_12']
    assert len(ab) == 2 and all(len(h) == 3 and all(len(c) == 2 for c in h) for h in ab)
    out = ctypes.create_string_buffer(384)
    assert _native.load().zkc_pairing_bin(ol.g1_json(mine['vk_alpha_1']), ol.g2_json(mine['vk_beta_2']), out) == 0
    assert [str(int.from_bytes(out.raw[32 * i:32 * i + 32], 'little')) for i in range(12)] == [x for h in ab for c in h for x in c]


def test_circuit_selection_by_wasm_hash():
    lib = _native.load()
    hexbuf = ctypes.create_string_buffer(65)
    junk = b'\0asm\x01\0\0\0' + bytes(100)
    assert lib.zkc_circuit_nlevels_from_wasm(junk, len(junk), hexbuf) == -1
    assert hexbuf.value.decode() == hashlib.sha256(junk).hexdigest()
    with pytest.raises(ValueError, match='unknown circuit wasm'):
        groth16.circuit_nlevels(junk)
    assert groth16.circuit_nlevels(None) == 160 and groth16.circuit_nlevels(None, 10) == 10
    d = ctypes.create_string_buffer(32); lib.zkc_sha256(b'abc', 3, d)
    assert d.raw == hashlib.sha256(b'abc').digest()
    # the hash table entry is the one the reference publishes for dev/160 (tests/golden/ref/circuits-info.md = circuits-info.md:7)
    info = open(ol.golden('ref/circuits-info.md')).read()
    sha = [l.split()[0] for l in info.splitlines() if l.strip().endswith('circuit.wasm')][0]
    assert sha == '80a73567f6a4655d4332301efcff4bc5711bb48176d1c71fdb1e48df222ac139'
    wasm = '/root/reference/artifacts/zkCensus/dev/160/circuit.wasm'
    if os.path.exists(wasm):                                      # only where the reference tree is present (not on the GPU box)
        raw = open(wasm, 'rb').read()
        assert lib.zkc_circuit_nlevels_from_wasm(raw, len(raw), hexbuf) == 160 and hexbuf.value.decode() == sha
        assert groth16.circuit_nlevels(wasm) == 160
        with pytest.raises(ValueError):
            groth16.circuit_nlevels(wasm, 10)


def test_verifier_strictness_and_return_codes():
    """Every return of the verifiers is 1, 0 or negative (a caller that tests rc > 0 must never accept on an error); JSON points need z in
    {0, 1}; B must be in the order-r subgroup; the verification key's points are checked too."""
    lib = _native.load()
    vk = ol.load_json('ref/verification_key.json'); pr = ol.load_json('ref/proof.json'); sig = ol.load_json('ref/signals.json')
    t = lambda x: json.dumps(x).encode()
    assert lib.zkc_verify(t(vk), t(sig), t(pr)) == 1
    bad = json.loads(json.dumps(pr)); bad['pi_a'][2] = '2'
    assert lib.zkc_verify(t(vk), t(sig), t(bad)) == 0
    bad = json.loads(json.dumps(pr)); bad['pi_b'][2] = ['1', '1']
    assert lib.zkc_verify(t(vk), t(sig), t(bad)) == 0
    badvk = json.loads(json.dumps(vk)); badvk['vk_alpha_1'][0] = str(int(badvk['vk_alpha_1'][0]) + 1)
    assert lib.zkc_verify(t(badvk), t(sig), t(pr)) < 0                                        # alpha not on the curve
    assert lib.zkc_verify(b'{}', t(sig), t(pr)) < 0 and lib.zkc_verify(None, t(sig), t(pr)) < 0
    # a twist point outside the order-r subgroup (oracle_lib.twist_point_outside_g2: a small x with a square right-hand side, Fq2 arithmetic in Python)
    pt = ol.twist_point_outside_g2()
    prb = bytearray(ol.proof_bytes(pr))
    prb[64:192] = b''.join(ol.le32(v) for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]))     # on the twist, (almost surely) not in the subgroup
    assert lib.zkc_verify_bin(ol.vk_bytes(vk), 8, b''.join(ol.le32(x) for x in sig), bytes(prb)) == 0
    # batch verifier without a usable context: negative, never positive
    rc = lib.zkc_verify_batch(None, ol.vk_bytes(vk), 8, b''.join(ol.le32(x) for x in sig), ol.proof_bytes(pr), 1, None)
    assert rc < 0


def test_random_scalars_are_field_elements():
    lib = _native.load()
    buf = ctypes.create_string_buffer(32 * 4000)
    import time
    t0 = time.perf_counter(); lib.zkc_random_scalars(buf, 4000); dt = time.perf_counter() - t0
    assert dt < 0.1, 'drawing 4000 scalars took %.0f ms: the blinding of a 1024-voter batch must not cost more than the proofs' % (dt * 1e3)
    v = [int.from_bytes(buf.raw[32 * i:32 * i + 32], 'little') for i in range(4000)]
    assert max(v) < ol.R and len(set(v)) == 4000
    assert max(v) > ol.R * 0.99 and sum(1 for x in v if x >> 248) > 3000        # uniform in Fr, not capped at 2^248 as in round 1


def test_groth16_prover_argument_checks_before_any_gpu_work(tmp_path):
    """rapidsnark's entry point: malformed files, witness-length mismatch and the SHORT_BUFFER size query are answered from the file headers alone
    (no context, no key load, no proof)."""
    from zkcensus_amd import setup
    lib = _native.load()
    _, zp, _ = setup.ensure_test_artifacts(10, directory=str(tmp_path))
    zk = open(zp, 'rb').read()
    nv = ol.lib().zko_n_wires(10)
    payload = bytes(32 * nv)
    need = lib.zkc_wtns_write(payload, nv, None, 0); w = ctypes.create_string_buffer(need); lib.zkc_wtns_write(payload, nv, w, need)
    err = ctypes.create_string_buffer(256)
    ps, us = ctypes.c_ulong(0), ctypes.c_ulong(0)
    rc = lib.groth16_prover(zk, len(zk), w.raw, need, None, ctypes.byref(ps), None, ctypes.byref(us), err, 256)
    assert rc == 2 and ps.value >= 768 and us.value >= 8 * 78 and b'too short' in err.value          # size query: no GPU needed here
    pb, ub = ctypes.create_string_buffer(ps.value), ctypes.create_string_buffer(us.value)
    short = ctypes.create_string_buffer(need - 32); lib.zkc_wtns_write(payload[32:], nv - 1, short, need - 32)
    rc = lib.groth16_prover(zk, len(zk), short.raw, need - 32, pb, ctypes.byref(ps), ub, ctypes.byref(us), err, 256)
    assert rc == 3 and b'Invalid witness length' in err.value
    rc = lib.groth16_prover(zk[:1000], 1000, w.raw, need, pb, ctypes.byref(ps), ub, ctypes.byref(us), err, 256)
    assert rc == 1
    rc = lib.groth16_prover(zk, len(zk), b'wtnsXXXX' + bytes(40), 48, pb, ctypes.byref(ps), ub, ctypes.byref(us), err, 256)
    assert rc == 1 and b'Invalid witness file' in err.value


def test_zkey_fingerprint_identifies_keys(tmp_path):
    """The per-call identity of the resident-key caches (groth16_prover, N-API, groth16.py): equal for equal images, different for two keys of the same
    circuit from different ceremonies, for a changed header byte and for a changed length; cheap (the full SHA-256 of a 55 MB key costs 0.27 s)."""
    import time
    from zkcensus_amd import setup
    lib = _native.load()
    _, za, _ = setup.ensure_test_artifacts(10, directory=str(tmp_path))
    _, zb, _ = setup.ensure_test_artifacts(10, seed=1234, directory=str(tmp_path))
    a, b = open(za, 'rb').read(), open(zb, 'rb').read()
    fp = lambda raw: (lambda o: (lib.zkc_zkey_fingerprint(raw, len(raw), o), o.raw))(ctypes.create_string_buffer(32))
    (ra, fa), (rb, fb) = fp(a), fp(b)
    assert ra == 0 and rb == 0 and fa != fb and fp(bytes(a))[1] == fa
    mut = bytearray(a); mut[12 + 12 + 4 + 12 + 100] ^= 1                    # a byte of alpha1 in the header section
    assert fp(bytes(mut))[1] != fa
    assert fp(a + b'\0')[1] != fa
    assert fp(b'not a zkey')[0] != 0
    t0 = time.perf_counter()
    for _ in range(20): fp(a)
    assert (time.perf_counter() - t0) / 20 < 0.02


def test_verifier_from_many_threads_with_the_key_cache():
    """[r5] zkc_verify keeps the latest verification keys ready by their bytes (csrc/zkc_verify.hip vk_ready): eight threads verify the reference's triple, a triple with a
    changed signal and triples under nine OTHER keys (the reference's with IC points permuted: well-formed keys under which the proof is invalid -- more keys than the cache
    holds, so entries are evicted while others use them) and every verdict is the sequential one."""
    from concurrent.futures import ThreadPoolExecutor
    lib = _native.load()
    vk = ol.load_json('ref/verification_key.json'); pr = ol.load_json('ref/proof.json'); sig = ol.load_json('ref/signals.json')
    t = lambda x: json.dumps(x).encode()
    keys = [vk]
    for i in range(9):
        k = json.loads(json.dumps(vk)); ic = k['IC']; a, b = 1 + i % 8, 1 + (i + 3) % 8
        if a == b: b = 1 + (b % 8)
        ic[a], ic[b] = ic[b], ic[a]; keys.append(k)
    bad_sig = list(sig); bad_sig[0] = str(int(sig[0]) + 1)
    jobs = []
    for rep in range(6):
        for ki, k in enumerate(keys):
            jobs.append((t(k), t(sig), t(pr), 1 if ki == 0 else 0))
            jobs.append((t(k), t(bad_sig), t(pr), 0))
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(lambda j: lib.zkc_verify(j[0], j[1], j[2]), jobs))
    assert got == [j[3] for j in jobs]
