"""GPU: the drop-in surfaces -- snarkjs-shaped groth16.fullProve / prove / verify (ts_inputs/src/example.ts:358-362) and
the rapidsnark-shaped groth16_prover C entry point (zk_census_test.go:89 via go-rapidsnark)."""
import ctypes, json
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_snarkjs_surface_and_rapidsnark_entry():
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import groth16, setup, _native
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(160)
    vk = json.load(open(vkey_path))
    ex = ol.load_json('ref/inputs_example.json')
    # groth16.fullProve(inputs, wasmFile, zkeyFile) -> {proof, publicSignals}
    out = groth16.fullProve(ex, '../artifacts/zkCensus/dev/160/circuit.wasm', zkey_path)
    assert out['publicSignals'] == ol.load_json('ref/signals.json')
    assert set(out['proof']) == {'pi_a', 'pi_b', 'pi_c', 'protocol', 'curve'} and out['proof']['pi_a'][2] == '1'
    assert groth16.verify(vk, out['publicSignals'], out['proof']) is True
    assert ol.verify(vk, out['publicSignals'], out['proof'])               # and the oracle agrees
    tampered = list(out['publicSignals']); tampered[2] = str(int(tampered[2]) ^ 1)
    assert groth16.verify(vk, tampered, out['proof']) is False
    # two proofs of the same statement differ (random r, s) but both verify; fixed (r, s) reproduces bytes
    out2 = groth16.fullProve(ex, None, zkey_path)
    assert out2['proof'] != out['proof'] and groth16.verify(vk, out2['publicSignals'], out2['proof'])
    a = groth16.fullProve(ex, None, zkey_path, rs=(5, 7)); b = groth16.fullProve(ex, None, zkey_path, rs=(5, 7))
    assert a == b
    # a failing circuit assert surfaces like snarkjs' "Assert Failed"
    bad = dict(ex); bad['voteWeight'] = str(int(ex['availableWeight']) + 1)
    with pytest.raises(RuntimeError, match='Assert Failed'):
        groth16.fullProve(bad, None, zkey_path)
    # rapidsnark entry point: file images in, JSON text out
    lib = _native.load()
    zk = open(zkey_path, 'rb').read()
    wt = groth16.wtns.calculate(ex)
    ps, us = ctypes.c_ulong(16), ctypes.c_ulong(16)
    err = ctypes.create_string_buffer(256)
    rc = lib.groth16_prover(zk, len(zk), wt, len(wt), ctypes.create_string_buffer(16), ctypes.byref(ps), ctypes.create_string_buffer(16), ctypes.byref(us), err, 256)
    assert rc == 2 and ps.value > 600                                       # PROVER_ERROR_SHORT_BUFFER with required sizes
    pb, ub = ctypes.create_string_buffer(ps.value), ctypes.create_string_buffer(us.value)
    rc = lib.groth16_prover(zk, len(zk), wt, len(wt), pb, ctypes.byref(ps), ub, ctypes.byref(us), err, 256)
    assert rc == 0, err.value
    assert json.loads(ub.value) == ol.load_json('ref/signals.json')
    assert groth16.verify(vk, json.loads(ub.value), json.loads(pb.value))
    short = wt[:-32 * 5]                                                     # truncated witness -> invalid file
    rc = lib.groth16_prover(zk, len(zk), short, len(short), pb, ctypes.byref(ps), ub, ctypes.byref(us), err, 256)
    assert rc == 1
