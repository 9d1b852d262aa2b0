"""CPU tests of the host logic around the hot path: the R1CS restatement of the circuit (satisfied by reference-wasm
witnesses, violated by mutations), the C-ABI export list, and the test-only setup at nLevels=10 (config 1 of
BASELINE.json: plumbing without a GPU) proved and verified by the oracle."""
import ctypes, json, os, random, sys
import pytest
import oracle_lib as ol
import zkcensus_amd
from zkcensus_amd import r1cs, setup, _native

sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
VEC = ol.load_json('witness_vectors.json')


@pytest.fixture(scope='module')
def circuit160():
    return r1cs.build(160)


def wires(w):
    return [int.from_bytes(w[32 * i:32 * i + 32], 'little') for i in range(len(w) // 32)]


def test_layout_matches_oracle_and_abi(circuit160):
    L, cs = circuit160
    lib = _native.load()
    for nl in (10, 31, 160, 252, 253):
        assert r1cs.Layout(nl).nWires == ol.lib().zko_n_wires(nl) == lib.zkc_circuit_n_wires(nl)
        assert r1cs.Layout(nl).nInputs == lib.zkc_circuit_n_inputs(nl)
    assert L.nWires == 82754 and len(cs.cons) + 9 <= 131072 and len(cs.cons) + 9 > 65536    # domain 2^17


def test_r1cs_satisfied_by_reference_witnesses(circuit160):
    L, cs = circuit160
    for v in VEC['vectors']:
        rc, w = ol.witness(v['inputs'])
        assert rc == 0 and cs.check(wires(w)) == -1, v['name']


def test_r1cs_rejects_mutated_witnesses(circuit160):
    L, cs = circuit160
    rc, w = ol.witness(VEC['vectors'][1]['inputs'])
    base = wires(w)
    rng = random.Random(5)
    # wires no constraint touches: voteHash (census.circom:54-57) -- every other wire must be pinned
    free = {4, 5}
    # ... and IsZero.inv hints whose input is zero (any value satisfies in*inv = 1 - out when in = 0, exactly as in circomlib),
    # plus the three inv wires of ForceEqualIfEnabled / isZero[nLevels] whose input is identically zero
    inp = VEC['vectors'][1]['inputs']
    for blk, sib in ((L.off_census, inp['censusSiblings']), (L.off_sikver, inp['sikSiblings'])):
        sib = list(sib) + ['0'] * (L.n - len(sib))
        oz = blk + L.off_iszero
        for i in range(L.n - 1):
            if int(sib[i]) == 0:
                free.add(oz + 2 * i + 1 if i < L.n - 2 else oz + 2 * (L.n - 2))
        free.add(blk + 2); free.add(oz + 2 * (L.n - 2) + 1)
    free.add(L.off_checknull)
    # the base witness satisfies every constraint (test_r1cs_satisfied_by_reference_witnesses), so a change of wire k can only break constraints that mention k: an index
    # wire -> constraints, instead of a walk over all ~130 k constraints per mutation (a minute of pure Python for 300 mutations)
    import collections
    from zkcensus_amd.r1cs import lc_eval
    touch = collections.defaultdict(list)
    for idx, (a, b, c) in enumerate(cs.cons):
        for wq in set(a) | set(b) | set(c):
            touch[wq].append(idx)
    violated = lambda m, k: any((lc_eval(cs.cons[i][0], m) * lc_eval(cs.cons[i][1], m) - lc_eval(cs.cons[i][2], m)) % ol.R for i in touch[k])
    for k in rng.sample(range(1, L.nWires), 300):
        if k in free:
            continue
        m = list(base); m[k] = (m[k] + 1 + rng.randrange(5)) % ol.R
        assert violated(m, k), 'wire %d is unconstrained' % k
    for k in (4, 5):
        m = list(base); m[k] = (m[k] + 1) % ol.R
        assert not violated(m, k) and cs.check(m) == -1


def test_abi_exports_every_declared_symbol():
    lib = _native.load()
    syms = _native.declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), 'libzkcensus.so does not export ' + s


def test_product_has_no_cpu_path():
    """Without a GPU the product must fail loudly, never fall back to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(zkcensus_amd.ZkcError) as ei:
        zkcensus_amd.Context(0)
    assert ei.value.code == 6
    src = open(os.path.join(ol.ROOT, 'zk-franchise-proof-circuit_amd', '__init__.py')).read()
    assert 'oracle' not in src.replace('the oracle', '')


def test_setup_prove_verify_nl10(tmp_path):
    from census_gen import random_voter
    rp, zp, vp = setup.ensure_test_artifacts(10, directory=str(tmp_path))
    zk = open(zp, 'rb').read(); vk = json.load(open(vp))
    z = ol.zkey_parse(zk)
    assert (z.nVars, z.nPublic, z.domainSize) == (8354, 8, 16384)
    assert ol.vk_bytes(vk) == ol.zkey_vk(zk)
    rng = random.Random(1)
    v = random_voter(rng, ol.poseidon, nLevels=10, depth_c=10, depth_s=4)
    rc, w = ol.witness(v, nLevels=10)
    assert rc == 0
    rc, proof, pub = ol.prove(zk, w, rng.randrange(ol.R), rng.randrange(ol.R))
    assert rc == 0 and ol.verify(vk, pub, proof)
    bad = bytearray(pub); bad[0] ^= 1
    assert not ol.verify(vk, bytes(bad), proof)
    # same seed -> same key (deterministic toxic waste); other seed -> other key
    _, zp2, _ = setup.ensure_test_artifacts(10, seed=7, directory=str(tmp_path))
    assert open(zp2, 'rb').read() != zk


def test_product_verifier_on_reference_triple():
    """zkc_verify (product, CPU) against the reference's committed proof / signals / verification key."""
    from zkcensus_amd import groth16
    vk = ol.load_json('ref/verification_key.json'); pr = ol.load_json('ref/proof.json'); sig = ol.load_json('ref/signals.json')
    assert groth16.verify(vk, sig, pr) is True
    assert groth16.verify(open(ol.golden('ref/verification_key.json')).read(), sig, open(ol.golden('ref/proof.json')).read()) is True
    for i in (0, 5, 7):
        s2 = list(sig); s2[i] = str((int(s2[i]) + 1) % ol.R)
        assert groth16.verify(vk, s2, pr) is False
    bad = json.loads(json.dumps(pr)); bad['pi_a'][0] = str((int(bad['pi_a'][0]) + 1) % ol.Q)
    assert groth16.verify(vk, sig, bad) is False
    bad = json.loads(json.dumps(pr)); bad['pi_b'][1][0] = str((int(bad['pi_b'][1][0]) + 1) % ol.Q)
    assert groth16.verify(vk, sig, bad) is False
    # binary form agrees with the oracle's verifier on the same inputs
    lib = _native.load()
    assert lib.zkc_verify_bin(ol.vk_bytes(vk), 8, b''.join(ol.le32(x) for x in sig), ol.proof_bytes(pr)) == 1


def test_codecs_roundtrip():
    lib = _native.load()
    from zkcensus_amd import groth16
    pr = ol.load_json('ref/proof.json'); sig = ol.load_json('ref/signals.json')
    pj, sj = groth16.proof_to_json(ol.proof_bytes(pr), b''.join(ol.le32(x) for x in sig))
    assert sj == sig and pj['pi_a'] == pr['pi_a'] and pj['pi_b'] == pr['pi_b'] and pj['pi_c'] == pr['pi_c']
    assert pj['protocol'] == 'groth16' and pj['curve'] == 'bn128'
    payload = b''.join(ol.le32(i * 7919 % ol.R) for i in range(100))
    need = lib.zkc_wtns_write(payload, 100, None, 0)
    buf = ctypes.create_string_buffer(need)
    assert lib.zkc_wtns_write(payload, 100, buf, need) == need and buf.raw[:4] == b'wtns'
    assert groth16._wtns_payload(buf.raw) == payload
    assert lib.zkc_wtns_parse(b'nope' + buf.raw[4:], need, None, None) != 0
    # short-buffer protocol of the rapidsnark-shaped surface
    ps, us = ctypes.c_ulong(8), ctypes.c_ulong(8)
    rc = lib.zkc_proof_to_json(ol.proof_bytes(pr), b''.join(ol.le32(x) for x in sig), 8, ctypes.create_string_buffer(8), ctypes.byref(ps),
                               ctypes.create_string_buffer(8), ctypes.byref(us))
    assert rc == 2 and ps.value >= 768 and us.value >= 8 * 78


def test_napi_shim_loads_and_verifies_reference_triple():
    """The Node N-API surface (napi/): groth16.verify on the reference triple, and fullProve failing loudly without a GPU."""
    import shutil, subprocess
    node = shutil.which('node')
    if not node or not os.path.exists(os.path.join(ol.ROOT, 'napi', 'zkcensus.node')):
        pytest.skip('node or the built addon is not available')
    js = ("const z=require('./napi');const vk=require('./tests/golden/ref/verification_key.json'),pr=require('./tests/golden/ref/proof.json'),"
          "sg=require('./tests/golden/ref/signals.json');(async()=>{const a=await z.groth16.verify(vk,sg,pr);sg[1]=(BigInt(sg[1])+1n).toString();"
          "const b=await z.groth16.verify(vk,sg,pr);console.log(JSON.stringify([a,b,z.flatten(require('./tests/golden/ref/inputs_example.json'),160).length]))})()")
    out = subprocess.check_output([node, '-e', js], cwd=ol.ROOT, timeout=120).decode()
    assert json.loads(out.strip().splitlines()[-1]) == [True, False, 334 * 32]
    # the snarkjs surface is complete (groth16.fullProve / prove / verify, wtns.calculate) and a wasm names the circuit by its sha256:
    # an unknown wasm is refused before any GPU work; the reference's dev/160 circuit.wasm (when the tree is here) selects nLevels = 160
    wasm = '/root/reference/artifacts/zkCensus/dev/160/circuit.wasm'
    js = ("const z=require('./napi');const inp=require('./tests/golden/ref/inputs_example.json');(async()=>{"
          "const api=[typeof z.groth16.fullProve,typeof z.groth16.prove,typeof z.groth16.verify,typeof z.wtns.calculate];"
          "let unknown=null;try{await z.groth16.fullProve(inp,Buffer.from('not a circuit'),'nokey.zkey')}catch(e){unknown=String(e)}"
          "let mism=null;try{await z.wtns.calculate(inp,%s,null,{nLevels:10})}catch(e){mism=String(e)}"
          "console.log(JSON.stringify({api,unknown,mism}))})()" % (json.dumps(wasm) if os.path.exists(wasm) else 'Buffer.from("x")'))
    r = json.loads(subprocess.check_output([node, '-e', js], cwd=ol.ROOT, timeout=120).decode().strip().splitlines()[-1])
    assert r['api'] == ['function'] * 4 and 'unknown circuit wasm' in r['unknown']
    assert ('nLevels=160 circuit but nLevels=10' in r['mism']) if os.path.exists(wasm) else ('unknown circuit wasm' in r['mism'])


def test_nlevels_253_the_largest_circuit_circomlib_permits():
    """SMTVerifier indexes a 254-bit Num2Bits_strict, so nLevels = 253 (realNLevels = 254) is the largest circuit that
    compiles.  There all 254 key bits steer a level and the bit Num2Bits' linear constraint solves for is the last level's: no reference wasm
    exists for this size, so the convention is this build's own -- what is checked is that the three restatements (R1CS generator, CPU oracle, and
    through the layout the GPU kernels) agree: the oracle's witnesses satisfy the R1CS, mutations do not, the domain is still 2^17."""
    L, cs = r1cs.build(253)
    assert L.nWires == ol.lib().zko_n_wires(253) == 128882            # 128 386 at nLevels = 252 plus one level per tree (2 x 249 wires) minus the two solved bits
    assert len(cs.cons) + 9 <= 131072
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    rng = random.Random(253)
    for dc, ds in ((9, 4), (253, 253), (254, 1)):
        v = random_voter(rng, ol.poseidon, nLevels=253, depth_c=dc, depth_s=ds)
        rc, w = ol.witness(v, 253)
        assert rc == 0, (dc, ds)
        ww = wires(w)
        assert cs.check(ww) == -1, (dc, ds)
        for _ in range(20):                                         # a flipped wire inside the tail of either verifier block breaks a constraint
            blk = rng.choice((L.off_census, L.off_sikver)); i = blk + L.lvl_off(L.n - 1) + rng.randrange(0, 12)
            m = list(ww); m[i] = (m[i] + 1) % ol.R
            assert cs.check(m) != -1, i
