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
    out = groth16.fullProve(ex, None, zkey_path)             # wasm None = the native nLevels = 160 circuit (a wasm would be matched by sha256)
    with pytest.raises(ValueError, match='unknown circuit wasm'):
        groth16.fullProve(ex, b'\0asm not the census circuit', zkey_path)
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
    # [r4] wasm None and no nLevels: the depth is read off the key (a voter of an nLevels-10 census through an nLevels-10 key, no option needed); a key that is no census key says so
    import random, sys, os
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    _, z10, v10 = setup.ensure_test_artifacts(10)
    assert groth16.key_nlevels(z10) == 10 and groth16.key_nlevels(zkey_path) == 160
    o10 = groth16.fullProve(random_voter(random.Random(3), ol.poseidon, nLevels=10, depth_c=4, depth_s=6), None, z10)
    assert groth16.verify(json.load(open(v10)), o10['publicSignals'], o10['proof'])
    # a failing circuit assert surfaces like snarkjs' "Assert Failed"
    bad = dict(ex); bad['voteWeight'] = str(int(ex['availableWeight']) + 1)
    with pytest.raises(RuntimeError) as e:
        groth16.fullProve(bad, None, zkey_path)
    assert str(e.value) == 'Assert Failed.\nError in template ZkFranchiseProofCircuit_234 line: 72\n'      # the wasm's own text (tests/golden "weight_exceeds")
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
    # re-entrant like rapidsnark's (goroutines call it concurrently): four threads, each proof verifies; then another key through the same
    # entry point (the resident key is identified by the sha256 of the file image, so it is replaced, not aliased) and back again
    import threading
    _, zkey10, vkey10 = setup.ensure_test_artifacts(10)
    zk10 = open(zkey10, 'rb').read(); vk10 = json.load(open(vkey10))
    import random, sys, os
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    v10 = random_voter(random.Random(3), ol.poseidon, nLevels=10, depth_c=5, depth_s=4)
    wt10 = groth16.wtns.calculate(v10, None, 10)
    results = []

    def call(zkb, wtb, key):
        p_s, u_s = ctypes.c_ulong(2048), ctypes.c_ulong(2048)
        p_b, u_b = ctypes.create_string_buffer(2048), ctypes.create_string_buffer(2048); e_b = ctypes.create_string_buffer(256)
        r = lib.groth16_prover(zkb, len(zkb), wtb, len(wtb), p_b, ctypes.byref(p_s), u_b, ctypes.byref(u_s), e_b, 256)
        results.append((r, key, p_b.value, u_b.value, e_b.value))
    th = [threading.Thread(target=call, args=(zk, wt, 160)) for _ in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    call(zk10, wt10, 10); call(zk, wt, 160); call(zk10, wt10, 10)
    assert len(results) == 7
    for r, key, pjs, ujs, e in results:
        assert r == 0, e
        assert groth16.verify(vk if key == 160 else vk10, json.loads(ujs), json.loads(pjs))
    assert len({pjs for _, _, pjs, _, _ in results}) == 7                    # fresh (r, s) every time


def test_key_cache_is_keyed_by_content(tmp_path):
    """groth16._key: a second, different key object (or a rewritten file at the same path) must never be served the first key's handle."""
    import torch  # noqa: F401
    from zkcensus_amd import groth16, setup
    _, z10, v10 = setup.ensure_test_artifacts(10)
    _, z10b, v10b = setup.ensure_test_artifacts(10, seed=99, directory=str(tmp_path))
    a, b = open(z10, 'rb').read(), open(z10b, 'rb').read()
    ka = groth16._key(a); kb = groth16._key(b)
    assert ka is not kb and groth16._key(bytes(a)) is ka and groth16._key(z10) is ka
    path = str(tmp_path / 'k.zkey')
    open(path, 'wb').write(a); k1 = groth16._key(path)
    open(path, 'wb').write(b); os_utime = __import__('os').utime; os_utime(path, None)
    k2 = groth16._key(path)
    assert k1 is ka and k2 is kb
    assert len(groth16._keys) <= groth16.MAX_RESIDENT_KEYS


def test_key_and_context_lifecycle_returns_device_memory():
    """Every load / prove / free cycle gives its HBM back (early-return leaks in zkc_zkey_load): free memory after ten cycles of
    (context, key, proof, batch of 70, malformed key, close) is where it was after the second."""
    import torch, random, numpy as np
    import zkcensus_amd
    from zkcensus_amd import setup
    import sys, os
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    _, zp, _ = setup.ensure_test_artifacts(10)
    zk = open(zp, 'rb').read()
    rng = random.Random(3)
    voters = [random_voter(rng, ol.poseidon, nLevels=10, depth_c=3, depth_s=2) for _ in range(70)]
    flat = b''.join(zkcensus_amd.flatten_inputs(v, 10) for v in voters)
    rs = b''.join(rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(140))
    torch.cuda.init(); torch.cuda.synchronize()

    def cycle():
        ctx = zkcensus_amd.Context(0)
        pk = zkcensus_amd.ProvingKey(ctx, zk)
        ws, st = ctx.witness(voters[:1], nLevels=10)
        pk.prove(ws[0], 3, 4)
        d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
        d_w = torch.empty(70 * ctx.n_wires(10) * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(70, dtype=torch.int32, device='cuda')
        pk.fullprove_batch_dev(d_in.data_ptr(), 70, d_w.data_ptr(), d_st.data_ptr(), rs)
        for cut in (len(zk) // 2, 4096, 64):                      # truncated images: the loader must release what it had allocated
            with pytest.raises(zkcensus_amd.ZkcError):
                zkcensus_amd.ProvingKey(ctx, zk[:cut])
        pk.close(); ctx.close()
        del d_in, d_w, d_st
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    cycle()                                                        # [r5] the device's lane streams (zkc_lane_streams) are made once per process and get their hardware queues --
    first = cycle()                                                # rings, scratch: ~0.5 GB -- on first use, which for some of them is the second cycle (tools/gpu/mem_cycles.py)
    for _ in range(8):
        last = cycle()
    assert first - last < 64 << 20, 'device memory shrank by %.1f MB over nine load/free cycles' % ((first - last) / 1e6)


def test_groth16_fullprove_takes_the_reference_argument_shapes():
    """[r5] prover.Prove(zkey, wasm, inputs) in one call (include/zkcensus.h groth16_fullprove; zk_census_test.go:81-93): file images and JSON text in, proof.json / public.json
    texts out, rapidsnark's return codes.  nLevels 10 key, circuit named by the key's own shape (wasm NULL); an unknown wasm is refused; a voter who fails a circuit assert and a
    damaged inputs object come back as 1 with the wasm's / circom_runtime's message; the size query is answered without proving."""
    import ctypes, json, random, sys, os
    import zkcensus_amd
    from zkcensus_amd import setup, _native
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    lib = _native.load()
    nl = 10
    _, zp, vp = setup.ensure_test_artifacts(nl)
    zk = open(zp, 'rb').read(); vk = json.load(open(vp))
    v = random_voter(random.Random(77), ol.poseidon, nLevels=nl, depth_c=5, depth_s=3)
    text = json.dumps(v).encode()

    def call(wasm, js, psz=2048, usz=2048):
        pb, ub, eb = ctypes.create_string_buffer(max(psz, 1)), ctypes.create_string_buffer(max(usz, 1)), ctypes.create_string_buffer(512)
        ps, us = ctypes.c_ulong(psz), ctypes.c_ulong(usz)
        rc = lib.groth16_fullprove(zk, len(zk), wasm, len(wasm) if wasm else 0, js, len(js), pb, ctypes.byref(ps), ub, ctypes.byref(us), eb, 512)
        return rc, pb.value, ub.value, eb.value.decode(), ps.value, us.value
    rc, pj, uj, err, ps, us = call(None, text)
    assert rc == 0, err
    proof, pub = json.loads(pj), json.loads(uj)
    rcw, w = ol.witness(v, nl); assert rcw == 0
    assert [int(x) for x in pub] == [int.from_bytes(w[32 * (1 + k):32 * (2 + k)], 'little') for k in range(8)]
    le = lambda x: int(x).to_bytes(32, 'little')
    pbin = le(proof['pi_a'][0]) + le(proof['pi_a'][1]) + le(proof['pi_b'][0][0]) + le(proof['pi_b'][0][1]) + le(proof['pi_b'][1][0]) + le(proof['pi_b'][1][1]) + le(proof['pi_c'][0]) + le(proof['pi_c'][1])
    assert ol.verify(vk, b''.join(le(x) for x in pub), pbin)
    assert ps == len(pj) + 1 and us == len(uj) + 1                      # sizes written back include the terminating NUL, as rapidsnark's do
    # size query: nothing proved, sizes that hold any proof of this shape
    rc, _, _, err, ps, us = call(None, text, 10, 10)
    assert rc == 2 and ps >= len(pj) + 1 and us >= len(uj) + 1
    # an unknown witness calculator: refused (the C ABI has no wasm runtime), with a message that says what to do
    rc, _, _, err, _, _ = call(b'\0asm\x01\0\0\0' + b'x' * 64, text)
    assert rc == 1 and 'no wasm runtime' in err
    # a voter who fails census.circom:72 (weight): the wasm's own message, nothing proved for him
    badv = dict(v, voteWeight=str(int(v['availableWeight']) + 1))
    rc, _, _, err, _, _ = call(None, json.dumps(badv).encode())
    assert rc == 1 and err.startswith('Assert Failed.'), err
    # a damaged document, an unknown signal
    rc, _, _, err, _, _ = call(None, text[:-5])
    assert rc == 1 and err.startswith('JSON'), err
    rc, _, _, err, _, _ = call(None, json.dumps(dict(v, extra='1')).encode())
    assert rc == 1 and err == 'Signal extra not found\n'
    # the binary form through an explicit service, with injected (r, s): bytes equal the oracle's
    svc = zkcensus_amd.ProvingService([0])
    p = ctypes.create_string_buffer(256); u = ctypes.create_string_buffer(256); st = ctypes.c_int32(0); eb = ctypes.create_string_buffer(256)
    rs = (12345).to_bytes(32, 'little') + (67890).to_bytes(32, 'little')
    rc = lib.zkc_service_fullprove_json(svc._h, zk, len(zk), None, 0, text, len(text), rs, p, u, ctypes.byref(st), eb, 256)
    assert rc == 0 and st.value == 0, eb.value
    rco, op, ou = ol.prove(zk, w, 12345, 67890)
    assert rco == 0 and (p.raw, u.raw) == (op, ou)
    svc.close()
