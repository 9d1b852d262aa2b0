"""The reference's own test file as a compiled client (tests/host/reference_test_shape.cc over include/zkcensus_prover.hpp): zk_census_test.go's getEnvVars and its three
tests -- Test_genInputs (internal.MockInputs -> inputs_example.json), Test_genProof (prover.Prove -> proof.Bytes() -> proof.json, signals.json), Test_verifyProof
(prover.ParseProof -> proof.Verify) -- with the same environment variables, artifacts tree and file names, run the way `make test` runs them (CIRCUIT_NAME / ENVIRONMENT /
NLEVELS in the environment, working directory = the repository root that holds ./artifacts).

CPU: Test_verifyProof and the Bytes() round trip on the reference's committed triple (tests/golden/ref).  GPU: all three tests in order at nLevels 10 and 160 under the
build's test keys, every artifact then checked by the oracle; failures surface as the reference's would.  Sorts early: the children start before this process touches the GPU."""
import json, os, re, shutil, subprocess
import pytest
import oracle_lib as ol

SRC = os.path.join(ol.ROOT, 'tests', 'host', 'reference_test_shape.cc')
LIBDIR = os.path.join(ol.ROOT, 'zk-franchise-proof-circuit_amd')
REF = os.path.join(ol.ROOT, 'tests', 'golden', 'ref')


def build(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, 'libzkcensus.so')):
        pytest.skip('libzkcensus.so is not built')
    exe = str(tmp_path / 'reference_test_shape')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-Wall', '-Wextra', '-Werror', SRC, '-I' + os.path.join(ol.ROOT, 'include'), '-L' + LIBDIR, '-lzkcensus',
                           '-Wl,-rpath,' + LIBDIR, '-o', exe])
    return exe


def run(exe, cwd, *args, **env):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    for k in ('CIRCUIT_NAME', 'ENVIRONMENT', 'NLEVELS', 'KEYSIZE', 'PADDING'):
        if k not in env:
            e.pop(k, None)
    try:
        return subprocess.run([exe, *args], cwd=str(cwd), env=e, capture_output=True, text=True, timeout=900)
    except OSError as err:
        pytest.skip('cannot start a child program from this process: %s' % err)


def tree(tmp_path, name, env, nl):
    d = tmp_path / 'artifacts' / name / env / str(nl)
    d.mkdir(parents=True)
    return d


def test_verify_and_bytes_on_the_reference_triple(tmp_path):
    """zk_census_test.go:103-124 on the files the reference commits (artifacts/zkCensus/dev/160: verification_key.json, proof.json, signals.json; defaults of getEnvVars)."""
    exe = build(tmp_path)
    d = tree(tmp_path, 'zkCensus', 'dev', 160)
    for f in ('verification_key.json', 'proof.json', 'signals.json'):
        shutil.copyfile(os.path.join(REF, f), d / f)
    r = run(exe, tmp_path, 'Test_verifyProof')
    assert r.returncode == 0 and '--- PASS: Test_verifyProof' in r.stdout, r.stdout + r.stderr
    r = run(exe, tmp_path, 'Test_bytesRoundTrip')                     # (*Proof).Bytes() writes exactly what the reference's run wrote
    assert r.returncode == 0, r.stdout + r.stderr
    # a signal changed: ParseProof still succeeds, Verify fails (the Go test's qt.Assert(err, qt.IsNil) would)
    sig = json.loads((d / 'signals.json').read_text()); sig[2] = str(int(sig[2]) + 1)
    (d / 'signals.json').write_text(json.dumps(sig, separators=(',', ':')))
    r = run(exe, tmp_path, 'Test_verifyProof')
    assert r.returncode == 1 and 'proof verification failed' in r.stdout
    # a document encoding/json refuses: ParseProof fails
    (d / 'proof.json').write_text(open(os.path.join(REF, 'proof.json')).read()[:-1])
    r = run(exe, tmp_path, 'Test_verifyProof')
    assert r.returncode == 1 and 'parsing proof' in r.stdout
    # internal/helpers.go:16-34 on example.ts:340-346's raw client values = the reference's inputs_example.json (the same voter)
    enc = json.loads(run(exe, tmp_path, 'Print_encodings').stdout)
    ref_inputs = json.load(open(os.path.join(REF, 'inputs_example.json')))
    assert enc == {k: ref_inputs[k] for k in ('electionId', 'voteHash', 'address', 'password', 'signature')}
    # getEnvVars: the reference's own refusals and its path scheme
    r = run(exe, tmp_path, 'Test_verifyProof', NLEVELS=9)
    assert r.returncode == 1 and 'the number of levels must be 10 at least to support the current key length' in r.stdout
    r = run(exe, tmp_path, 'Test_verifyProof', NLEVELS=16, KEYSIZE=3)
    assert r.returncode == 1 and 'the key size can not be bigger than ceil(nLevels/8)' in r.stdout
    r = run(exe, tmp_path, 'Test_verifyProof', CIRCUIT_NAME='other', ENVIRONMENT='stage', NLEVELS=250)
    assert r.returncode == 1 and './artifacts/other/stage/250/verification_key.json' in r.stdout
    r = run(exe, tmp_path, 'Test_verifyProof', NLEVELS='abc')           # strconv.Atoi fails: the default stays
    assert './artifacts/zkCensus/dev/160/' in r.stdout or 'parsing proof' in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize('nl', [10, 160])
def test_the_three_reference_tests_in_order(tmp_path, nl):
    from zkcensus_amd import setup
    exe = build(tmp_path)
    _, zp, vp = setup.ensure_test_artifacts(nl)
    d = tree(tmp_path, 'zkCensus', 'dev', nl)
    shutil.copyfile(zp, d / "proving_key.zkey"); shutil.copyfile(vp, d / "verification_key.json")
    r = run(exe, tmp_path, NLEVELS=nl)
    assert r.returncode == 0, r.stdout + r.stderr
    assert [l for l in r.stdout.splitlines() if l.startswith('---')] == ['--- PASS: Test_genInputs', '--- PASS: Test_genProof', '--- PASS: Test_verifyProof']
    text = (d / 'inputs_example.json').read_text()
    inputs = json.loads(text)
    # internal/inputs.go:14-31, 82-97: member names and order, weights 10 / 5, the election of :57, sibling lists of nLevels + 1 ending in "0"
    assert list(inputs) == ['electionId', 'nullifier', 'availableWeight', 'voteHash', 'sikRoot', 'censusRoot', 'address', 'password', 'signature', 'voteWeight',
                            'censusSiblings', 'sikSiblings']
    ref_inputs = json.load(open(os.path.join(REF, 'inputs_example.json')))
    assert inputs['electionId'] == ref_inputs['electionId'] and inputs['voteHash'] == ref_inputs['voteHash'] and inputs['password'] == ref_inputs['password']
    assert (inputs['availableWeight'], inputs['voteWeight']) == ('10', '5')
    assert len(inputs['censusSiblings']) == len(inputs['sikSiblings']) == nl + 1 and inputs['censusSiblings'][-1] == inputs['sikSiblings'][-1] == '0'
    assert 1 <= sum(s != '0' for s in inputs['censusSiblings']) <= 10                       # a ten-leaf tree
    if nl == 160:                                                                            # json.MarshalIndent(inputs, "", "\t"): the fixture's own layout
        assert re.sub(r'\d+', 'N', text) == re.sub(r'\d+', 'N', open(os.path.join(REF, 'inputs_example.json')).read().rstrip('\n'))
    # the oracle on what the three tests wrote: the circuit accepts the generated voter, the public signals are its wires 1..8, the proof verifies
    rc, w = ol.witness(inputs, nl); assert rc == 0
    pub = json.loads((d / 'signals.json').read_text()); proof = json.loads((d / 'proof.json').read_text())
    assert [int(x) for x in pub] == [int.from_bytes(w[32 * (1 + k):32 * (2 + k)], 'little') for k in range(8)]
    assert list(proof) == ['pi_a', 'pi_b', 'pi_c']                                          # proof.Bytes(): ProofData only, as the committed proof.json
    assert re.sub(r'\d+', 'N', (d / 'proof.json').read_text()) == re.sub(r'\d+', 'N', open(os.path.join(REF, 'proof.json')).read())
    le = lambda x: int(x).to_bytes(32, 'little')
    pbin = b''.join(le(x) for x in (proof['pi_a'][0], proof['pi_a'][1], proof['pi_b'][0][0], proof['pi_b'][0][1], proof['pi_b'][1][0], proof['pi_b'][1][1], proof['pi_c'][0], proof['pi_c'][1]))
    assert ol.verify(json.load(open(vp)), b''.join(le(x) for x in pub), pbin)
    r = run(exe, tmp_path, 'Test_bytesRoundTrip', NLEVELS=nl)
    assert r.returncode == 0, r.stdout
    # a voter who spends more than he has (census.circom:72): prover.Prove returns the witness calculator's error, nothing is written
    os.remove(d / 'proof.json')
    (d / 'inputs_example.json').write_text(json.dumps(dict(inputs, voteWeight='11')))
    r = run(exe, tmp_path, 'Test_genProof', NLEVELS=nl)
    assert r.returncode == 1 and 'Assert Failed.' in r.stdout and not (d / 'proof.json').exists()
