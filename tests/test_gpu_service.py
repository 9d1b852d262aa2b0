"""GPU: the submission queue behind the single-proof entry points (include/zkcensus.h zkc_service_*, csrc/zkc_service.hip).

The reference proves one voter per call -- prover.Prove per voter, from goroutines (zk_census_test.go:89, ending in rapidsnark's groth16_prover) and
groth16.fullProve per ballot (ts_inputs/src/example.ts:358-362).  These tests call the same way from many threads and check that every caller gets
ITS proof (bytes equal the oracle's for its own (witness, r, s)), that callers were really coalesced into shared pipeline passes, that a voter who
fails a circuit assert fails alone, and that the rapidsnark entry point sustains the batch regime from 64 concurrent callers."""
import ctypes, json, os, random, sys, threading, time
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))


def _voters(n, nl, seed, **kw):
    from census_gen import random_voter
    rng = random.Random(seed)
    return [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(2, 8), depth_s=rng.randrange(2, 8), **kw) for _ in range(n)]


def test_concurrent_callers_share_passes_and_get_their_own_proofs():
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import setup
    nl, T, per = 10, 48, 3
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(nl)
    zk = open(zkey_path, 'rb').read(); vk = json.load(open(vkey_path))
    voters = _voters(T * per, nl, 11)
    bad = 7                                                             # this caller's ballot claims more weight than the census gives it
    voters[bad] = dict(voters[bad], voteWeight=str(int(voters[bad]['availableWeight']) + 1))
    rng = random.Random(5)
    rs = [rng.randrange(ol.R).to_bytes(32, 'little') + rng.randrange(ol.R).to_bytes(32, 'little') for _ in voters]
    svc = zkcensus_amd.ProvingService([0])
    out = [None] * len(voters)

    def caller(t):
        for k in range(per):
            i = t * per + k
            out[i] = svc.fullprove(zk, voters[i], nLevels=nl, rs=rs[i])
    th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
    for t in th: t.start()
    for t in th: t.join()
    st = svc.stats()
    assert st['requests'] == T * per and st['failed'] == 0 and st['waiting'] == 0
    assert st['batches'] < st['requests'] // 2 and st['largest_batch'] >= 8, st        # coalesced: far fewer batches than callers
    assert st['key_loads'] == 1 and st['devices_used'] == 1
    def check(i):
        proof, pub, status = out[i]
        if i == bad:
            assert status == 1                                          # ZKC_W_ERR_WEIGHT (census.circom:72); nobody else was affected
            return
        assert status == 0
        rc, w = ol.witness(voters[i], nl); assert rc == 0
        rc, oproof, opub = ol.prove(zk, w, int.from_bytes(rs[i][:32], 'little'), int.from_bytes(rs[i][32:], 'little'))
        assert rc == 0 and proof == oproof and pub == opub, 'caller %d got a proof that is not the oracle\'s for its own inputs' % i
        if i % 16 == 0:
            assert ol.verify(vk, pub, proof)
    ol.pmap(check, range(len(out)))
    # the witness path (groth16.prove shape): same queue, host witnesses
    rc, w0 = ol.witness(voters[0], nl)
    p, u = svc.prove(zk, w0, rs=rs[0])
    assert (p, u) == (out[0][0], out[0][1])
    with pytest.raises(zkcensus_amd.ZkcError) as e:
        svc.prove(zk, w0[:-64])
    assert e.value.code == 3                                            # INVALID_WITNESS_LENGTH, that caller alone
    # a second key through the same service becomes resident BESIDE the first ([r4] several keys per device) and both keep producing valid proofs
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        _, z2, v2 = setup.ensure_test_artifacts(nl, seed=77, directory=d)
        zk2 = open(z2, 'rb').read(); vk2 = json.load(open(v2))
        p2, u2, s2 = svc.fullprove(zk2, voters[1], nLevels=nl)
        assert s2 == 0 and ol.verify(vk2, u2, p2) and not ol.verify(vk, u2, p2)
        p1, u1, s1 = svc.fullprove(zk, voters[1], nLevels=nl)
        assert s1 == 0 and ol.verify(vk, u1, p1)
    assert svc.stats()['key_loads'] == 2                                # going back to the first key did not reload it
    svc.close()


def test_rapidsnark_entry_point_from_64_threads_nl160():
    """groth16_prover (the symbol go-rapidsnark binds, zk_census_test.go:89) called from 64 threads with 64 different witnesses: every proof verifies, the
    callers were coalesced, and the rate is that of the batch regime, not 64 x the single-proof latency."""
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import groth16, setup, _native
    nl, T, per = 160, 64, 4
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(nl)
    zk = open(zkey_path, 'rb').read(); vk = json.load(open(vkey_path))
    ctx = zkcensus_amd.Context(0)
    from census_gen import random_voter
    rng = random.Random(21)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(10, 18), depth_s=rng.randrange(10, 18)) for _ in range(T)]
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0] * T
    lib = _native.load()
    images = []
    for w in ws:
        n = lib.zkc_wtns_write(w, len(w) // 32, None, 0); buf = ctypes.create_string_buffer(n); lib.zkc_wtns_write(w, len(w) // 32, buf, n); images.append(buf.raw)
    ctx.close()
    results = [[] for _ in range(T)]

    def caller(t):
        for _ in range(per):
            ps, us = ctypes.c_ulong(2048), ctypes.c_ulong(2048)
            pb, ub, eb = ctypes.create_string_buffer(2048), ctypes.create_string_buffer(2048), ctypes.create_string_buffer(256)
            rc = lib.groth16_prover(zk, len(zk), images[t], len(images[t]), pb, ctypes.byref(ps), ub, ctypes.byref(us), eb, 256)
            results[t].append((rc, pb.value, ub.value, eb.value))
    caller(0); results[0].clear()                                       # key load and work-space growth are not part of the rate
    svc = zkcensus_amd.ProvingService(default=True)
    before = svc.stats()
    th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    after = svc.stats()
    n = T * per
    for t in range(T):
        assert len(results[t]) == per
        for rc, pjs, ujs, e in results[t]:
            assert rc == 0, e
            assert [int(x) for x in json.loads(ujs)] == [int.from_bytes(ws[t][32 * (1 + k):32 * (2 + k)], 'little') for k in range(8)]
    sample = [(t, k) for t in range(0, T, 7) for k in range(per)]
    for t, k in sample:
        assert groth16.verify(vk, json.loads(results[t][k][2]), json.loads(results[t][k][1]))
    assert len({results[t][k][1] for t in range(T) for k in range(per)}) == n           # fresh (r, s) per call
    batches = after['batches'] - before['batches']
    rate = n / dt
    print('\ngroth16_prover x %d threads: %d proofs in %.3f s = %.0f proofs/s, %d batches (largest %d)' % (T, n, dt, rate, batches, after['largest_batch']))
    assert batches <= n // 4, (before, after)
    assert rate > 800, 'rapidsnark entry point from %d threads: %.0f proofs/s' % (T, rate)      # r02's global mutex: ~250 proofs/s; measured here: see profiles/r03_service_*.json


def test_failed_work_space_growth_leaves_the_key_usable(monkeypatch):
    """a failed hipMalloc inside lanes_ensure used to leave freed buffers behind a stale capacity.  ZKC_TEST_FAIL_ALLOC makes the growth to
    >= N proofs in flight fail: the call must return an error, and the next smaller call must re-allocate and prove."""
    import torch, numpy as np
    import zkcensus_amd
    from zkcensus_amd import setup
    nl = 10
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(nl)
    zk = open(zkey_path, 'rb').read(); vk = json.load(open(vkey_path))
    voters = _voters(12, nl, 4)
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    rs = b''.join(random.Random(9).randrange(ol.R).to_bytes(32, 'little') for _ in range(24))
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    d_w = torch.empty(12 * ctx.n_wires(nl) * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(12, dtype=torch.int32, device='cuda')
    p2, u2 = pk.fullprove_batch_dev(d_in.data_ptr(), 2, d_w.data_ptr(), d_st.data_ptr(), rs[:128])          # work space for 2 in flight
    monkeypatch.setenv('ZKC_TEST_FAIL_ALLOC', '8')
    with pytest.raises(zkcensus_amd.ZkcError) as e:
        pk.fullprove_batch_dev(d_in.data_ptr(), 12, d_w.data_ptr(), d_st.data_ptr(), rs)                     # growth to 12 fails
    assert e.value.code == 6 and 'ZKC_TEST_FAIL_ALLOC' in str(e.value)
    p2b, u2b = pk.fullprove_batch_dev(d_in.data_ptr(), 2, d_w.data_ptr(), d_st.data_ptr(), rs[:128])        # re-allocates for 2 and proves
    assert (p2b, u2b) == (p2, u2)
    monkeypatch.delenv('ZKC_TEST_FAIL_ALLOC')
    p12, u12 = pk.fullprove_batch_dev(d_in.data_ptr(), 12, d_w.data_ptr(), d_st.data_ptr(), rs)
    assert p12[:512] == p2 and ol.verify(vk, u12[-256:], p12[-256:])
    pk.close(); ctx.close()


def test_async_submit_and_destroy_with_requests_pending():
    """zkc_service_submit_fullprove (what the N-API addon calls): completions arrive by callback on a service thread; destroying the service while requests are still
    queued completes every one of them exactly once -- proved, or failed with 'shut down' -- and never crashes or leaks a caller waiting."""
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import setup, _native
    nl, N = 10, 200
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(nl)
    zk = open(zkey_path, 'rb').read(); vk = json.load(open(vkey_path))
    voters = _voters(8, nl, 31)
    flats = [zkcensus_amd.flatten_inputs(v, nl) for v in voters]
    lib = _native.load()
    svc = zkcensus_amd.ProvingService([0])
    svc.fullprove(zk, flats[0], nLevels=nl)                             # key resident before the burst
    DONE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_char_p)
    proofs = [ctypes.create_string_buffer(256) for _ in range(N)]; pubs = [ctypes.create_string_buffer(256) for _ in range(N)]
    seen = []; lock = threading.Lock()

    def done(user, rc, status, text):
        with lock:
            seen.append((int(user or 0), rc, status, (text or b'').decode()))
    cb = DONE(done)
    for i in range(N):
        rc = lib.zkc_service_submit_fullprove(svc._h, zk, len(zk), nl, flats[i % 8], None, ctypes.cast(proofs[i], ctypes.c_char_p), ctypes.cast(pubs[i], ctypes.c_char_p),
                                              ctypes.cast(cb, ctypes.c_void_p), ctypes.c_void_p(i + 1))
        assert rc == 0
    svc.close()                                                         # destroy with most of them still queued or in flight
    assert sorted(u for u, *_ in seen) == list(range(1, N + 1))         # every request completed exactly once
    ok = [u for u, rc, st, _ in seen if rc == 0]; down = [u for u, rc, st, t in seen if rc != 0]
    assert all('shut down' in t for u, rc, st, t in seen if rc != 0)
    assert len(ok) + len(down) == N and len(ok) >= 1
    for u in ok[:20]:
        assert ol.verify(vk, pubs[u - 1].raw, proofs[u - 1].raw)
    # submitting to a bad key image fails at once, nothing is queued
    svc2 = zkcensus_amd.ProvingService([0])
    assert lib.zkc_service_submit_fullprove(svc2._h, b'zkeyXXXXXXXXXXXX', 16, nl, flats[0], None, ctypes.cast(proofs[0], ctypes.c_char_p), None, ctypes.cast(cb, ctypes.c_void_p), None) == 5
    assert svc2.stats()['requests'] == 0
    svc2.close()


def test_queue_spills_over_further_device_entries(monkeypatch):
    """Several device entries (device 0 listed three times stands in for three GPUs on a one-GPU box: three contexts, three resident keys, six workers): with a low spill
    threshold a loaded queue brings the other entries up -- each loads the key from the service's own copy of the image BEFORE it takes requests -- and every caller
    still gets the proof of its own inputs."""
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import setup
    nl, T, per = 10, 48, 6
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(nl)
    zk = open(zkey_path, 'rb').read(); vk = json.load(open(vkey_path))
    voters = _voters(T * per, nl, 41)
    rng = random.Random(6)
    rs = [rng.randrange(ol.R).to_bytes(32, 'little') + rng.randrange(ol.R).to_bytes(32, 'little') for _ in voters]
    monkeypatch.setenv('ZKC_SERVICE_SPILL', '6')
    svc = zkcensus_amd.ProvingService([0, 0, 0])
    monkeypatch.delenv('ZKC_SERVICE_SPILL')
    out = [None] * len(voters)

    def caller(t):
        for k in range(per):
            i = t * per + k
            out[i] = svc.fullprove(bytes(zk) if t % 2 else zk, voters[i], nLevels=nl, rs=rs[i])          # two different caller buffers holding the same image: one key
    th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
    for t in th: t.start()
    for t in th: t.join()
    st = svc.stats()
    assert st['requests'] == T * per and st['failed'] == 0 and st['waiting'] == 0 and st['devices'] == 3
    assert 1 <= st['key_loads'] <= 3 and st['devices_used'] >= 1
    def check(i):
        rc, w = ol.witness(voters[i], nl)
        rc2, op, ou = ol.prove(zk, w, int.from_bytes(rs[i][:32], 'little'), int.from_bytes(rs[i][32:], 'little'))
        if not (rc == 0 and rc2 == 0 and out[i] == (op, ou, 0)):
            whose = [j for j in range(len(voters)) if out[j][0] == out[i][0]]
            raise AssertionError('caller %d: rc %d/%d, status %r, public signals %s, proof %s; verifier on what it got: %r; callers holding these bytes: %r' % (
                i, rc, rc2, out[i][2], 'equal' if out[i][1] == ou else 'DIFFER', 'equal' if out[i][0] == op else 'DIFFERS', ol.verify(vk, out[i][1], out[i][0]), whose))
    ol.pmap(check, range(0, len(voters), 13))
    # an image that differs from the resident one ONLY in bytes the sampled fingerprint does not look at is another key: it must not be served the resident one
    import struct
    lib = svc._lib
    fp_a = ctypes.create_string_buffer(32); fp_b = ctypes.create_string_buffer(32)
    twin = bytearray(zk); pos = None
    for cand in range(len(zk) // 2, len(zk) // 2 + 200000, 37):                     # flip a byte until the fingerprint stays the same (most bytes are unsampled)
        t2 = bytearray(zk); t2[cand] ^= 1
        lib.zkc_zkey_fingerprint(zk, len(zk), fp_a); lib.zkc_zkey_fingerprint(bytes(t2), len(t2), fp_b)
        if fp_a.raw == fp_b.raw:
            twin, pos = t2, cand; break
    assert pos is not None
    loads_before = svc.stats()['key_loads']
    try:
        p, u, s = svc.fullprove(bytes(twin), voters[0], nLevels=nl, rs=rs[0])       # a corrupted key: either refused by the loader or proves something that is NOT the good key's proof
        assert (p, u) != out[0][:2] or svc.stats()['key_loads'] > loads_before
    except zkcensus_amd.ZkcError:
        pass
    assert svc.stats()['key_loads'] > loads_before                                 # it was treated as a different key (loaded, or load attempted), never aliased
    svc.close()


@pytest.mark.parametrize('keys_per_device', [4, 1])
def test_two_keys_hammered_concurrently(monkeypatch, keys_per_device):
    """[r4] VERDICT r3 item 2.  The reference keeps a key per environment and per depth (circuit/circuit-compiler.sh:15,82); here two keys -- nLevels 10 and nLevels 160 -- are
    hammered by 32 threads EACH at the same time on one GPU.  With room for both (ZKC_SERVICE_KEYS >= 2, the default is 4) each is loaded once and never again; with room for
    one the service has to switch under load -- every switch frees a key while the other worker may be mid-call, the interleaving round 3's use-after-free needed -- and must
    neither fault nor hand anybody a proof that is not the one for ITS key, inputs and (r, s)."""
    import torch, numpy as np
    import zkcensus_amd
    from zkcensus_amd import setup
    from census_gen import random_voter
    T, per = 32, 2
    keys = {}
    for nl in (10, 160):
        _, zp, vp = setup.ensure_test_artifacts(nl)
        keys[nl] = (open(zp, 'rb').read(), json.load(open(vp)))
    rng = random.Random(77 + keys_per_device)
    voters = {10: _voters(T * per, 10, 5), 160: [random_voter(rng, ol.poseidon, nLevels=160, depth_c=rng.randrange(8, 16), depth_s=rng.randrange(8, 16)) for _ in range(T * per)]}
    rs = {nl: [rng.randrange(ol.R).to_bytes(32, 'little') + rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(T * per)] for nl in (10, 160)}
    monkeypatch.setenv('ZKC_SERVICE_KEYS', str(keys_per_device))
    svc = zkcensus_amd.ProvingService([0])
    monkeypatch.delenv('ZKC_SERVICE_KEYS')
    out = {10: [None] * (T * per), 160: [None] * (T * per)}
    errors = []

    def caller(nl, t):
        try:
            for k in range(per):
                i = t * per + k
                out[nl][i] = svc.fullprove(keys[nl][0], voters[nl][i], nLevels=nl, rs=rs[nl][i])
        except Exception as e:          # noqa: BLE001 -- reported below, with the thread that saw it
            errors.append((nl, t, repr(e)))
    th = [threading.Thread(target=caller, args=(nl, t)) for t in range(T) for nl in (10, 160)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors[:3]
    st = svc.stats(); ev = svc.timing()['key_evictions']
    assert st['requests'] == 2 * T * per and st['failed'] == 0 and st['waiting'] == 0
    if keys_per_device >= 2:
        assert st['key_loads'] == 2 and ev == 0, st                     # both keys resident side by side: zero loads after the first two
    else:
        assert st['key_loads'] >= 2 and ev == st['key_loads'] - 1, (st, ev)      # one slot: every load but the first evicted the other key
    print('\n[two keys, %d slot(s)] %d requests in %d batches, %d key loads, %d evictions' % (keys_per_device, st['requests'], st['batches'], st['key_loads'], ev))
    svc.close()
    # nLevels 10: every proof is the oracle's for that caller's inputs and (r, s)
    def check10(i):
        proof, pub, status = out[10][i]
        assert status == 0
        rc, w = ol.witness(voters[10][i], 10); assert rc == 0
        rc, oproof, opub = ol.prove(keys[10][0], w, int.from_bytes(rs[10][i][:32], 'little'), int.from_bytes(rs[10][i][32:], 'little'))
        assert rc == 0 and (proof, pub) == (oproof, opub), 'nLevels 10, caller %d' % i
    # ... beside the ONE nLevels-160 proof the oracle re-makes (15 s on a core): the last caller's, from the oracle's own witness
    i160 = T * per - 1

    def oracle160(_):
        rc, w = ol.witness(voters[160][i160], 160); assert rc == 0
        rc, oproof, opub = ol.prove(keys[160][0], w, int.from_bytes(rs[160][i160][:32], 'little'), int.from_bytes(rs[160][i160][32:], 'little'))
        assert rc == 0 and (oproof, opub) == out[160][i160][:2], 'nLevels 160: the service\'s proof is not the oracle\'s'
    from concurrent.futures import ThreadPoolExecutor
    bg = ThreadPoolExecutor(1); fut160 = bg.submit(oracle160, None)
    ol.pmap(check10, range(T * per) if keys_per_device == 1 else range(0, T * per, 2))          # every proof in the switching run, every other one where nothing switched
    # nLevels 160: every proof equals, byte for byte, the single-context batch path's for the same inputs and (r, s) (which tests/test_gpu_prover.py pins to the oracle's bytes),
    # the oracle's verifier accepts a sample under the key's verification key, and one of them is re-proved by the oracle itself (started above)
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, keys[160][0])
    B = T * per
    flat = b''.join(zkcensus_amd.flatten_inputs(v, 160) for v in voters[160])
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    d_w = torch.empty(B * ctx.n_wires(160) * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
    proofs, pubs = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), b''.join(rs[160]))
    assert int(d_st.abs().sum().item()) == 0
    for i, (proof, pub, status) in enumerate(out[160]):
        assert status == 0 and proof == proofs[256 * i:256 * i + 256] and pub == pubs[256 * i:256 * i + 256], 'nLevels 160, caller %d' % i
    for i in range(0, B, 9):
        assert ol.verify(keys[160][1], out[160][i][1], out[160][i][0])
    fut160.result(); bg.shutdown()
    pk.close(); ctx.close()


def test_soak_of_mixed_batch_sizes_on_two_keys(capsys):
    """[r4] tools/service_soak.py for four seconds at nLevels 10: 32 callers with random think times on two keys, inputs and witness calls mixed, so that batches of one, two, a
    few and dozens of proofs follow each other on both call slots -- the load under which round 3's shared-scratch race produced a wrong proof.  Every proof is verified under the
    key it was asked for and carries the public signals of the voter it was asked for (25 s at nLevels 160: profiles/r04_service_soak.json, 41 638 proofs in 3 679 batches)."""
    import service_soak
    old = sys.argv
    service_soak.SECONDS, service_soak.T, service_soak.NL = 4.0, 32, 10
    try:
        rc = service_soak.main()
    finally:
        sys.argv = old
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc == 0 and out['all_valid'] and out['proofs'] > 500 and out['batches'] > 50 and out['key_loads'] == 2, out


def test_six_keys_cycling_are_loaded_once_each_and_memory_is_reported(monkeypatch):
    """[r5] ADVICE r4 (medium, twice) + VERDICT r4 item 8.  Six keys of the nLevels-10 circuit (six ceremonies: the reference keeps a key per environment and depth,
    circuit/circuit-compiler.sh:15,82) cycle through one device that may hold eight: round 4 kept at most FOUR key images whatever the devices held, so the fifth key pushed a
    resident key's image out and its next caller loaded it a second time beside an unreachable copy.  Now: key_loads == number of keys however long they cycle, every proof
    valid under ITS key, and zkc_service_memory says what the keys hold.  Then the same service is told its device is out of memory while it holds >= 3 keys
    (ZKC_TEST_FAIL_KEY_LOADS): the next new key evicts idle keys until its load succeeds instead of failing the batch."""
    import tempfile
    import zkcensus_amd
    from zkcensus_amd import setup
    nl, NK = 10, 6
    monkeypatch.setenv('ZKC_SERVICE_KEYS', '8')
    svc = zkcensus_amd.ProvingService([0])
    monkeypatch.delenv('ZKC_SERVICE_KEYS')
    keys = []
    with tempfile.TemporaryDirectory() as d:
        for k in range(NK + 2):
            _, zp, vp = setup.ensure_test_artifacts(nl, seed=1000 + k, directory=os.path.join(d, str(k)))
            keys.append((open(zp, 'rb').read(), json.load(open(vp))))
    voters = _voters(8, nl, 3)
    out = []
    for rnd in range(4):                                                 # four rounds over six keys, several callers per key and round
        def caller(j):
            k = j % NK
            p, u, s = svc.fullprove(keys[k][0], voters[j % 8], nLevels=nl)
            assert s == 0
            out.append((k, p, u))
        th = [threading.Thread(target=caller, args=(j,)) for j in range(3 * NK)]
        for t in th: t.start()
        for t in th: t.join()
        assert svc.stats()['key_loads'] == NK, (rnd, svc.stats())      # each key once, in the first round; never again
    assert svc.timing()['key_evictions'] == 0
    for k, p, u in out[::5]:
        assert ol.verify(keys[k][1], u, p) and not ol.verify(keys[(k + 1) % NK][1], u, p)
    mem = svc.memory()
    assert mem['resident_keys'] == NK and mem['table_bytes'] > 0 and mem['work_bytes'] > 0 and mem['reserve_failures'] == 0, mem
    # an nLevels-10 key: ~0.1 GB of tables (the figure of the FIRST key of a device also holds what the device context allocates once, e.g. the hardware queues' rings:
    # the numbers are hipMemGetInfo deltas around the load), lanes' work space dominated by the 65 536 buckets per H job whatever the circuit's size
    assert mem['largest_key_table_bytes'] < 2e9 and mem['largest_device_work_bytes'] < 8e9 and mem['work_bytes'] == mem['largest_device_work_bytes'] and mem['table_bytes'] < 3e9, mem      # ONE work space for the six keys
    # out of memory with idle keys resident: evict and retry
    monkeypatch.setenv('ZKC_TEST_FAIL_KEY_LOADS', '3')
    p, u, s = svc.fullprove(keys[NK][0], voters[0], nLevels=nl)
    monkeypatch.delenv('ZKC_TEST_FAIL_KEY_LOADS')
    assert s == 0 and ol.verify(keys[NK][1], u, p)
    st = svc.stats(); ev = svc.timing()['key_evictions']
    assert st['failed'] == 0 and ev >= NK - 2 and svc.memory()['resident_keys'] <= 3, (st, ev, svc.memory())
    # ... and the survivors still prove (a key that was evicted is loaded again, once)
    p, u, s = svc.fullprove(keys[0][0], voters[1], nLevels=nl)
    assert s == 0 and ol.verify(keys[0][1], u, p)
    svc.close()


def test_four_nl160_keys_stay_under_the_stated_bound(monkeypatch):
    """[r5] VERDICT r4 item 8: what a resident key costs, measured where the driver sees it.  Four nLevels-160 keys (four ceremonies) on one device with the service's defaults
    (4 lanes x 64-proof passes): each key's constant tables are ~2.5 GB (3.1 for the first, which also pays the device context's one-time allocations); the lanes' work space
    (~37 GB) belongs to the device's context and is shared by its keys.  INTEGRATION.md section 1 states the bound: 4 keys <= 4 x 3.5 + 40 GB = 54 GB of the card's 288
    (158 GB while every key owned its lanes)."""
    import tempfile
    import zkcensus_amd
    from zkcensus_amd import setup
    from census_gen import random_voter
    nl = 160
    svc = zkcensus_amd.ProvingService([0])
    rng = random.Random(8)
    v = random_voter(rng, ol.poseidon, nLevels=nl, depth_c=12, depth_s=11)
    seen = []
    r1, zp0, vp0 = setup.ensure_test_artifacts(nl)                       # the circuit's .r1cs is the same for every ceremony: only the setup runs again (~8 s of C++ each)
    lib = zkcensus_amd._native.load()
    with tempfile.TemporaryDirectory() as d:
        for k in range(4):
            zp, vp = (zp0, vp0) if k == 0 else (os.path.join(d, '%d.zkey' % k), os.path.join(d, '%d.vkey.json' % k))
            if k:
                err = ctypes.create_string_buffer(512)
                assert lib.zkc_setup_from_r1cs(r1.encode(), 500 + k, zp.encode(), vp.encode(), err, 512) == 0, err.value
            zk = open(zp, 'rb').read(); vk = json.load(open(vp))
            p, u, s = svc.fullprove(zk, v, nLevels=nl)
            assert s == 0 and ol.verify(vk, u, p)
            seen.append(svc.memory())
            del zk
    mem = seen[-1]
    print('\n[four nLevels-160 keys] ' + '; '.join('%d keys: tables %.1f GB, work %.1f GB' % (m['resident_keys'], m['table_bytes'] / 1e9, m['work_bytes'] / 1e9) for m in seen))
    assert mem['resident_keys'] == 4 and svc.stats()['key_loads'] == 4 and svc.timing()['key_evictions'] == 0
    assert 2.5e9 < mem['largest_key_table_bytes'] < 3.5e9, mem
    assert mem['work_bytes'] < 40e9 and seen[0]['work_bytes'] == mem['work_bytes'] and mem['table_bytes'] + mem['work_bytes'] < 55e9, mem      # the lanes' work space is the device's: the second to fourth key added tables only
    svc.close()


def test_host_side_pass_layout_falls_back_for_witnesses_and_inputs_it_cannot_trust():
    """[r5] The service lays a one-pass call out from depths it reads on the HOST -- the sibling lists of the inputs, or the sibling wires of a witness computed elsewhere
    (csrc/zkc_service.hip depths_of; prove_batch_begin host_depths) -- so that begin never waits for the GPU.  The shortcut must be invisible:
      * a FOREIGN witness that does not carry the voter-independent template below its own sibling depth (a tampered wire deep in a tree, a tampered old-key block) is proved
        again from its fold flags (prove_batch_finish) and gives the ORACLE's bytes for that witness -- the oracle never folds -- while its neighbours in the same batch, which
        do fold, give theirs;
      * a voter whose LAST sibling is not zero (SMTLevIns, status 5 / 7) has no depth: its call takes the usual path, it is rejected alone, its neighbours are proved."""
    import zkcensus_amd
    from zkcensus_amd import setup
    nl = 10
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(nl)
    zk = open(zkey_path, 'rb').read()
    voters = _voters(12, nl, 19)
    rng = random.Random(23)
    rs = [rng.randrange(ol.R).to_bytes(32, 'little') + rng.randrange(ol.R).to_bytes(32, 'little') for _ in voters]
    wit = []
    for v in voters:
        rc, w = ol.witness(v, nl); assert rc == 0; wit.append(bytearray(w))
    L_census = 13 + 2 * nl                                               # first wire of the census verifier block (zkc_device.h WitnessLayout)
    nW = len(wit[0]) // 32
    lvl9 = 261 + 244 + 8 * 245 + 1                                       # WitnessLayout::lvl_off(9) at n = 11: level 9 of a tree, far below every voter's leaf (depths 2-7)
    wit[3][32 * (L_census + lvl9 + 20):32 * (L_census + lvl9 + 21)] = (12345).to_bytes(32, 'little')      # a hash-internal wire of that level: differs from the template there
    wit[7][32 * (nW - 40):32 * (nW - 40) + 32] = (777).to_bytes(32, 'little')                          # the sik verifier's tail (not a foldable group): no fold flag changes
    lib = zkcensus_amd._native.load(); retries0 = lib.zkc_debug_early_retries()
    svc = zkcensus_amd.ProvingService([0])
    out = [None] * len(voters)

    def caller(i):
        out[i] = svc.prove(zk, bytes(wit[i]), rs=rs[i])
    th = [threading.Thread(target=caller, args=(i,)) for i in range(len(voters))]
    for t in th: t.start()
    for t in th: t.join()
    assert svc.stats()['failed'] == 0
    assert lib.zkc_debug_early_retries() > retries0, 'the tampered witness should have been refused by the fold check and proved again'

    def check(i):
        rc, p, u = ol.prove(zk, bytes(wit[i]), int.from_bytes(rs[i][:32], 'little'), int.from_bytes(rs[i][32:], 'little'))
        assert rc == 0 and out[i] == (p, u), 'witness %d' % i
    ol.pmap(check, range(len(voters)))
    vk = json.load(open(vkey_path))
    assert ol.verify(vk, out[0][1], out[0][0]) and not ol.verify(vk, out[3][1], out[3][0])         # the tampered witness is no witness: its proof is the oracle's and does not verify
    # inputs path: a voter with a non-zero last sibling among good ones
    bad = dict(voters[5]); sib = list(bad['censusSiblings']); sib[nl] = '5'; bad['censusSiblings'] = sib
    group = [voters[0], bad, voters[1], voters[2]]
    res = [None] * 4

    def fcaller(i):
        res[i] = svc.fullprove(zk, group[i], nLevels=nl, rs=rs[i])
    th = [threading.Thread(target=fcaller, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert res[1][2] == 5 and [r[2] for i, r in enumerate(res) if i != 1] == [0, 0, 0]
    for i, v in ((0, voters[0]), (2, voters[1]), (3, voters[2])):
        rc, w = ol.witness(v, nl)
        rc, p, u = ol.prove(zk, w, int.from_bytes(rs[i][:32], 'little'), int.from_bytes(rs[i][32:], 'little'))
        assert (res[i][0], res[i][1]) == (p, u), i
    svc.close()
