"""GPU, Node: the reference's host language.  `node napi/example.js <zkey> <vkey>` is the ts_inputs/src/example.ts:358-362 call through the N-API
addon.  This file sorts first on purpose: node is started as a child process BEFORE this pytest process has initialised the GPU (the GPU
boxes refuse an exec from a process that already has)."""
import json
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_node_addon_full_surface_on_gpu(tmp_path):
    """The reference's host language: node napi/example.js <zkey> <vkey> = the ts_inputs/src/example.ts:358-362 call through the N-API addon on the
    GPU -- fullProve, wtns.calculate + prove with injected (r, s), four concurrent fullProve calls, a batch over a two-entry device pool, a failing assert, a buffer that is no wasm,
    [r4] a wasm without a native circuit (executed in Node, proved on the GPU) and a key of another depth with wasmFile null."""
    import os, shutil, subprocess
    from zkcensus_amd import setup
    node = shutil.which('node')
    addon = os.path.join(ol.ROOT, 'napi', 'zkcensus.node')
    if not node or not os.path.exists(addon):
        pytest.skip('node or the built addon is not available on this box')
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(160)
    # 64 different voters for the Promise.all burst (one fullProve per voter, the reference's call shape under load)
    import random, sys
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    rng = random.Random(64)
    voters_path = str(tmp_path / 'voters.json')
    json.dump([random_voter(rng, ol.poseidon, nLevels=160, depth_c=rng.randrange(10, 18), depth_s=rng.randrange(10, 18)) for _ in range(64)], open(voters_path, 'w'))
    # [r4] the wasm fallback: a key for the circuit of tests/golden/toy_passthrough.wasm (wires [1, out, a, b]: a . 1 = out ; (a + b) . 1 = out + b), made by the test-only setup;
    # and an nLevels-10 census key + voter for "wasmFile null: the depth comes from the key"
    import ctypes
    from zkcensus_amd import r1cs, _native
    cs = r1cs.R1CS(4, 1); cs.add({2: 1}, {0: 1}, {1: 1}); cs.add({2: 1, 3: 1}, {0: 1}, {1: 1, 3: 1})
    toy_r1cs, toy_zkey, toy_vkey = str(tmp_path / 'toy.r1cs'), str(tmp_path / 'toy.zkey'), str(tmp_path / 'toy_vkey.json')
    cs.write(toy_r1cs)
    err = ctypes.create_string_buffer(512)
    assert _native.load().zkc_setup_from_r1cs(toy_r1cs.encode(), 20241004, toy_zkey.encode(), toy_vkey.encode(), err, 512) == 0, err.value
    _, zkey10, vkey10 = setup.ensure_test_artifacts(10)
    voter10_path = str(tmp_path / 'voter10.json')
    json.dump(random_voter(random.Random(10), ol.poseidon, nLevels=10, depth_c=5, depth_s=3), open(voter10_path, 'w'))
    try:
        r = subprocess.run([node, os.path.join(ol.ROOT, 'napi', 'example.js'), zkey_path, vkey_path, '-', voters_path, toy_zkey, toy_vkey, zkey10, vkey10, voter10_path],
                           cwd=ol.ROOT, capture_output=True, text=True, timeout=900)
    except OSError as e:                                   # the box refused to start a child program from this process
        pytest.skip('cannot start node from this process: %s' % e)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j['verified'] is True and j['publicSignals'] == ol.load_json('ref/signals.json')
    assert j['wasmFallback'] and all(j['wasmFallback'].values()), j['wasmFallback']        # unknown wasm -> executed in Node -> GPU proof verifies; asserts carry the wasm's own text
    assert j['depthFromKey'] is True
    assert j['twoStepEqual'] and j['concurrentOk'] and j['batchOk'] and j['badInputRejected'] and j['unknownWasmRejected']
    b = j['burst']
    print('\nPromise.all over %d fullProve: %.1f ms = %d proofs/s; over %d: %d proofs/s' % (b['voters'], b['ms'], b['proofsPerSec'], b['voters4x'], b['proofsPerSec4x']))
    assert b['allVerified'] and b['signalsOk']
    assert b['proofsPerSec4x'] > 800, b            # r02 (one proof per libuv work item behind a mutex): ~130 proofs/s
