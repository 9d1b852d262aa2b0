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
    for nl in (10, 31, 160, 252):
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
    for k in rng.sample(range(1, L.nWires), 300):
        if k in free:
            continue
        m = list(base); m[k] = (m[k] + 1 + rng.randrange(5)) % ol.R
        assert cs.check(m) != -1, 'wire %d is unconstrained' % k
    for k in free:
        m = list(base); m[k] = (m[k] + 1) % ol.R
        assert cs.check(m) == -1


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
