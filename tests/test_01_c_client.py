"""A plain C99 program as the client of include/zkcensus.h (tests/host/abi_client.c): what a cgo preamble or any C host sees -- no C++, no torch,
no Python in the process that proves.  CPU: the header compiles as strict C (-std=c99 -pedantic -Werror) and the entry points resolve.  GPU: the
program proves voters through a device pool and this test checks its output with the oracle.  Sorts early on purpose: the child is started before
this pytest process has initialised the GPU (the GPU boxes refuse to start programs from a process that has)."""
import json, os, random, subprocess, sys
import pytest
import oracle_lib as ol

SRC = os.path.join(ol.ROOT, 'tests', 'host', 'abi_client.c')
LIB = os.path.join(ol.ROOT, 'zk-franchise-proof-circuit_amd', 'libzkcensus.so')


def build(tmp_path):
    exe = str(tmp_path / 'abi_client')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', SRC, '-I' + os.path.join(ol.ROOT, 'include'), '-ldl', '-o', exe])
    return exe


def test_header_is_c99_and_symbols_resolve(tmp_path):
    if not os.path.exists(LIB):
        pytest.skip('libzkcensus.so is not built')
    out = subprocess.check_output([build(tmp_path), LIB], timeout=120)
    assert json.loads(out) == {'header_compiles_as_c': True, 'symbols_resolve': True}


@pytest.mark.gpu
def test_c_program_proves_through_a_device_pool(tmp_path):
    import zkcensus_amd
    from zkcensus_amd import setup
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    nl, n = 10, 7
    _, zp, vp = setup.ensure_test_artifacts(nl)
    rng = random.Random(41)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(1, nl), depth_s=rng.randint(1, nl)) for _ in range(n)]
    voters[3] = dict(voters[3]); voters[3]['nullifier'] = str(int(voters[3]['nullifier']) ^ 1)          # census.circom:114
    inp = tmp_path / 'inputs.bin'; inp.write_bytes(b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters))
    outp = tmp_path / 'proofs.bin'
    try:
        r = subprocess.run([build(tmp_path), LIB, zp, str(inp), str(n), str(nl), str(outp)], capture_output=True, text=True, timeout=600)
    except OSError as e:
        pytest.skip('cannot start a child program from this process: %s' % e)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {'voters': n, 'rc': 7, 'failed_asserts': 1}      # ZKC_ERR_WITNESS: voter 3, the others are proved
    blob = outp.read_bytes(); proofs, pubs = blob[:256 * n], blob[256 * n:]
    vk = json.load(open(vp))
    for i in range(n):
        if i != 3:
            assert ol.verify(vk, pubs[256 * i:256 * i + 256], proofs[256 * i:256 * i + 256]), i


@pytest.mark.gpu
def test_c_program_proves_from_the_reference_inputs_file_image(tmp_path):
    """[r5] prover.Prove(zkey, wasm, inputs []byte) as the reference calls it (zk_census_test.go:81-93): the C program hands groth16_fullprove the FILE IMAGE of
    tests/golden/ref/inputs_example.json (the reference's own fixture) and of the nLevels-160 test key, and gets proof.json / public.json texts back -- which must carry the
    reference's own public signals (signals.json) and pass the pinned verifier under the key's verification key."""
    from zkcensus_amd import setup
    _, zp, vp = setup.ensure_test_artifacts(160)
    pj, uj = tmp_path / 'proof.json', tmp_path / 'public.json'
    try:
        r = subprocess.run([build(tmp_path), LIB, 'json', zp, os.path.join(ol.ROOT, 'tests', 'golden', 'ref', 'inputs_example.json'), str(pj), str(uj)], capture_output=True, text=True, timeout=600)
    except OSError as e:
        pytest.skip('cannot start a child program from this process: %s' % e)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {'rc': 0, 'size_query_rc': 2, 'damaged_inputs_rc': 1, 'damaged_inputs_message_is_set': True}
    proof = json.load(open(pj)); pub = json.load(open(uj))
    assert pub == json.load(open(os.path.join(ol.ROOT, 'tests', 'golden', 'ref', 'signals.json')))
    assert proof['protocol'] == 'groth16' and proof['curve'] == 'bn128'
    le = lambda x: int(x).to_bytes(32, 'little')
    pb = le(proof['pi_a'][0]) + le(proof['pi_a'][1]) + le(proof['pi_b'][0][0]) + le(proof['pi_b'][0][1]) + le(proof['pi_b'][1][0]) + le(proof['pi_b'][1][1]) + le(proof['pi_c'][0]) + le(proof['pi_c'][1])
    assert ol.verify(json.load(open(vp)), b''.join(le(x) for x in pub), pb)
