"""GPU parity of the proving stages (through the C ABI) against the CPU oracle: buildABC, NTT/joinABC, every MSM,
and whole proofs with injected (r, s) -- identical bytes, and accepted by the oracle's pairing verifier."""
import json, os, random, sys
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
R = ol.R
RINV = pow(1 << 256, -1, R)


@pytest.fixture(scope='module')
def env():
    import torch, zkcensus_amd
    from zkcensus_amd import setup
    ctx = zkcensus_amd.Context(0)
    keys = {}

    def get(nl):
        if nl not in keys:
            _, zp, vp = setup.ensure_test_artifacts(nl)
            zk = open(zp, 'rb').read()
            keys[nl] = (zk, zkcensus_amd.ProvingKey(ctx, zk), json.load(open(vp)))
        return keys[nl]
    yield ctx, get, torch
    for _, pk, _ in keys.values():
        pk.close()
    ctx.close()


def dev_bytes(torch, b):
    import numpy as np
    return torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy()).cuda()


def voter_witness(nl, seed):
    from census_gen import random_voter
    rng = random.Random(seed)
    v = random_voter(rng, ol.poseidon, nLevels=nl, depth_c=min(nl, 7), depth_s=min(nl, 5))
    rc, w = ol.witness(v, nLevels=nl)
    assert rc == 0
    return v, w


def test_build_abc_and_h_evals_nl10(env):
    ctx, get, torch = env
    zk, pk, _ = get(10)
    _, w = voter_witness(10, 1)
    dw = dev_bytes(torch, w)
    n = pk.domain_size
    got = pk.debug_stage(dw.data_ptr(), 0)
    A, B, C = ol.build_abc(zk, w)
    def demont(b):   # Montgomery bytes -> standard ints
        return [int.from_bytes(b[32 * i:32 * i + 32], 'little') * RINV % R for i in range(len(b) // 32)]
    ga, gb, gc = demont(got[:32 * n]), demont(got[32 * n:64 * n]), demont(got[64 * n:])
    std = lambda b: [int.from_bytes(b[32 * i:32 * i + 32], 'little') for i in range(n)]
    assert ga == std(A) and gb == std(B) and gc == std(C)
    assert pk.debug_stage(dw.data_ptr(), 1) == ol.h_evals(zk, w)


def test_each_msm_nl10(env):
    ctx, get, torch = env
    zk, pk, _ = get(10)
    z = ol.zkey_parse(zk)
    import ctypes
    rng = random.Random(3)
    _, w = voter_witness(10, 2)
    sections = [(0, z.pointsA, pk.n_vars, 64), (1, z.pointsB1, pk.n_vars, 64), (2, z.pointsB2, pk.n_vars, 128),
                (3, z.pointsC, pk.n_vars - pk.n_public - 1, 64), (4, z.pointsH, pk.domain_size, 64)]
    q, qinv = ol.Q, pow(1 << 256, -1, ol.Q)
    for which, ptr, cnt, psz in sections:
        raw = ctypes.string_at(ptr, cnt * psz)         # Montgomery coordinates as stored in the zkey
        std = b''.join((int.from_bytes(raw[32 * i:32 * i + 32], 'little') * qinv % q).to_bytes(32, 'little') for i in range(len(raw) // 32))
        # scalar mix: random field elements, witness-like small values, zeros, r-1
        sc = [rng.randrange(R) for _ in range(cnt)]
        for i in range(0, cnt, 7): sc[i] = rng.choice([0, 1, 2, R - 1, (1 << 128) - 1])
        for i in range(cnt // 2, cnt // 2 + 600): sc[i % cnt] = 1          # a heavy bucket
        scb = b''.join(x.to_bytes(32, 'little') for x in sc)
        got = pk.msm_debug(which, dev_bytes(torch, scb).data_ptr(), cnt)
        exp = ol.msm_g2(std, scb) if which == 2 else ol.msm_g1(std, scb)
        assert got == exp, 'MSM section %d' % which


def test_msm_equal_and_opposite_bases_nl10(env):
    """Wires that share a base point (same polynomial in the key) land in one bucket when their digits agree: the accumulation
    then has to double a point, or cancel P + (-P), instead of adding.  The radix-2^29 kernel detects both from the carried
    difference; sections A (G1) and B2 (G2) of the test key hold groups of up to 127 identical points."""
    ctx, get, torch = env
    zk, pk, _ = get(10)
    z = ol.zkey_parse(zk)
    import ctypes, collections
    q, qinv = ol.Q, pow(1 << 256, -1, ol.Q)
    rng = random.Random(11)
    for which, ptr, psz in ((0, z.pointsA, 64), (1, z.pointsB1, 64), (2, z.pointsB2, 128)):
        cnt = pk.n_vars
        raw = ctypes.string_at(ptr, cnt * psz)
        std = b''.join((int.from_bytes(raw[32 * i:32 * i + 32], 'little') * qinv % q).to_bytes(32, 'little') for i in range(len(raw) // 32))
        groups = collections.defaultdict(list)
        for i in range(cnt):
            pt = raw[psz * i:psz * i + psz]
            if any(pt): groups[pt].append(i)
        dup = max(groups.values(), key=len)
        assert len(dup) >= 3, 'test key lost its duplicated base points'
        a, b, c = dup[:3]
        cases = [{a: 3, b: 3}, {a: 3, b: (1 << 13) - 3}, {a: 5, b: 5, c: 5}, {a: 7, b: R - 7}, {a: 1, b: 1, c: R - 1},
                 {i: 9 for i in dup},                                         # a whole group in one bucket (doubling, then additions of 2P + P ...)
                 {i: (9 if k % 2 else (1 << 13) - 9) for k, i in enumerate(dup)}]
        base = [rng.randrange(R) for _ in range(cnt)]
        cases.append(dict(enumerate(base)) | {i: base[dup[0]] for i in dup})   # random everywhere, the group shares one full-size scalar
        for vals in cases:
            sc = [0] * cnt
            for k, v in vals.items(): sc[k] = v
            scb = b''.join(x.to_bytes(32, 'little') for x in sc)
            got = pk.msm_debug(which, dev_bytes(torch, scb).data_ptr(), cnt)
            exp = ol.msm_g2(std, scb) if which == 2 else ol.msm_g1(std, scb)
            assert got == exp, 'MSM section %d with equal bases' % which


@pytest.mark.parametrize('nl', [10, 160])
def test_prove_matches_oracle_and_verifies(env, nl):
    ctx, get, torch = env
    zk, pk, vk = get(nl)
    rng = random.Random(nl)
    for seed in (11, 12):
        _, w = voter_witness(nl, seed)
        r, s = rng.randrange(R), rng.randrange(R)
        proof, pub = pk.prove(w, r, s)
        assert pub == w[32:32 * 9]
        assert ol.verify(vk, pub, proof)
        if nl == 10 or seed == 11:
            rc, oproof, opub = ol.prove(zk, w, r, s)
            assert rc == 0 and oproof == proof and opub == pub
        bad = bytearray(proof); bad[70] ^= 4
        assert not ol.verify(vk, pub, bytes(bad))


def test_example_voter_end_to_end(env):
    ctx, get, torch = env
    zk, pk, vk = get(160)
    ex = ol.load_json('ref/inputs_example.json')
    ws, st = ctx.witness([ex])
    assert st == [0]
    proof, pub = pk.prove(ws[0], 0x1234567, 0x7654321)
    assert [str(int.from_bytes(pub[32 * i:32 * i + 32], 'little')) for i in range(8)] == ol.load_json('ref/signals.json')
    assert ol.verify(vk, pub, proof)
    # determinism: same (zkey, wtns, r, s) -> same bytes
    assert pk.prove(ws[0], 0x1234567, 0x7654321)[0] == proof
    # invalid witness length is reported like snarkjs / rapidsnark do
    import zkcensus_amd
    with pytest.raises(zkcensus_amd.ZkcError) as ei:
        pk.prove(ws[0][:-32], 1, 2)
    assert ei.value.code == 3 and 'Invalid witness length' in str(ei.value)


def test_batch_prove_with_folding_and_without(env):
    """zkc_prove_batch_dev: voters of different depths share pipeline passes (constant folding on); a witness that does not
    match the template (random field elements) must take the unfolded path -- both must equal the oracle's bytes."""
    ctx, get, torch = env
    from census_gen import random_voter
    nl = 10
    zk, pk, vk = get(nl)
    rng = random.Random(99)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=d % (nl + 1), depth_s=(3 * d) % (nl + 1)) for d in range(11)]
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0] * len(voters)
    junk = b''.join([(1).to_bytes(32, 'little')] + [rng.randrange(R).to_bytes(32, 'little') for _ in range(pk.n_vars - 1)])
    ws = ws + [junk]
    B = len(ws)
    rs = [(rng.randrange(R), rng.randrange(R)) for _ in range(B)]
    rsb = b''.join(r.to_bytes(32, 'little') + s.to_bytes(32, 'little') for r, s in rs)
    dw = dev_bytes(torch, b''.join(ws))
    proofs, pubs = pk.prove_batch_dev(dw.data_ptr(), B, rsb)
    for i in range(B):
        rc, op, opub = ol.prove(zk, ws[i], rs[i][0], rs[i][1])
        assert rc == 0 and op == proofs[256 * i:256 * i + 256], 'proof %d' % i
        assert opub == pubs[256 * i:256 * i + 256]
        if i < B - 1:
            assert ol.verify(vk, opub, op)


def test_batch_prove_nl160_verifies(env):
    ctx, get, torch = env
    from zkcensus_amd import census
    zk, pk, vk = get(160)
    voters = census.synthetic_census(ctx, 20)
    ws, st = ctx.witness(voters)
    assert st == [0] * 20
    rng = random.Random(160)
    rsb = b''.join(rng.randrange(R).to_bytes(32, 'little') for _ in range(40))
    proofs, pubs = pk.prove_batch_dev(dev_bytes(torch, b''.join(ws)).data_ptr(), 20, rsb)
    for i in range(20):
        assert ol.verify(vk, pubs[256 * i:256 * i + 256], proofs[256 * i:256 * i + 256]), i
    # one of them against the oracle prover, byte for byte
    rc, op, _ = ol.prove(zk, ws[7], int.from_bytes(rsb[64 * 7:64 * 7 + 32], 'little'), int.from_bytes(rsb[64 * 7 + 32:64 * 7 + 64], 'little'))
    assert rc == 0 and op == proofs[256 * 7:256 * 8]


def test_max_levels_nl253_config5(env):
    """the circuit at nLevels = 253, the largest circomlib permits (SMTVerifier indexes a 254-bit Num2Bits_strict): 128 882 wires,
    domain 2^17.  All 254 key bits steer a level there, so the bit Num2Bits solves for is the last level's (tests/test_r1cs_setup_cpu.py).
    Witness and proof bytes equal the oracle's, the pinned verifier accepts -- for a shallow voter (most levels fold) and for a voter at
    the maximum depth (nothing folds)."""
    ctx, get, torch = env
    nl = 253
    zk, pk, vk = get(nl)
    assert pk.n_vars == ol.lib().zko_n_wires(nl) == 128882 and pk.domain_size == 1 << 17
    from census_gen import random_voter
    rng = random.Random(5)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=7, depth_s=5), random_voter(rng, ol.poseidon, nLevels=nl, depth_c=nl, depth_s=nl)]
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0, 0]
    gpu = [pk.prove(w, 111, 222) for w in ws]

    def check(k):
        rc, ow = ol.witness(voters[k], nLevels=nl)
        assert rc == 0 and ws[k] == ow
        rc, op, opub = ol.prove(zk, ws[k], 111, 222)
        assert rc == 0 and gpu[k] == (op, opub)
        assert ol.verify(vk, gpu[k][1], gpu[k][0])
    ol.pmap(check, range(2))


def test_fullprove_batch_equals_witness_then_prove(env):
    """zkc_fullprove_batch_dev pipelines witness generation under the MSMs; bytes must equal the two-step path (and so the oracle's),
    statuses must report a voter whose inputs fail a circuit assert, and that voter must not disturb its neighbours."""
    ctx, get, torch = env
    import numpy as np, zkcensus_amd
    nl, B = 10, 150                                     # 3 passes of 64: chunk boundaries inside the batch
    zk, pk, vk = get(nl)
    from census_gen import random_voter
    rng = random.Random(77)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(1, nl), depth_s=rng.randint(1, nl)) for _ in range(B)]
    bad = 70
    voters[bad] = dict(voters[bad]); voters[bad]['nullifier'] = str(int(voters[bad]['nullifier']) ^ 1)      # census.circom:111 assert
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = dev_bytes(torch, flat)
    nW = ctx.n_wires(nl)
    d_w = torch.zeros(B * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
    rs = b''.join(rng.randrange(1 << 248).to_bytes(32, 'little') for _ in range(2 * B))
    p1, pub1 = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    st = d_st.cpu().tolist()
    assert st[bad] != 0 and all(x == 0 for i, x in enumerate(st) if i != bad)
    ws, st2 = ctx.witness(voters, nLevels=nl)
    assert st2 == st
    got_w = d_w.cpu().numpy().tobytes()
    for i in (0, 63, 64, 127, 128, B - 1):
        assert got_w[i * nW * 32:(i + 1) * nW * 32] == ws[i]
    d_w2 = dev_bytes(torch, b''.join(ws))
    p2, pub2 = pk.prove_batch_dev(d_w2.data_ptr(), B, rs)
    good = [i for i in range(B) if i != bad]
    assert all(p1[256 * i:256 * i + 256] == p2[256 * i:256 * i + 256] and pub1[256 * i:256 * i + 256] == pub2[256 * i:256 * i + 256] for i in good)
    for i in (0, 64, B - 1):
        assert ol.verify(vk, pub1[256 * i:256 * i + 256], p1[256 * i:256 * i + 256])
        r_i = int.from_bytes(rs[64 * i:64 * i + 32], 'little'); s_i = int.from_bytes(rs[64 * i + 32:64 * i + 64], 'little')
        rc, op, opub = ol.prove(zk, ws[i], r_i, s_i)
        assert rc == 0 and op == p1[256 * i:256 * i + 256]


def test_optional_pipeline_modes_give_identical_proofs(env):
    """ZKC_LANES=2 (two pipeline lanes taking alternate passes) and a small ZKC_INFLIGHT (many short passes) are scheduling options only:
    same bytes as the default configuration."""
    ctx, get, torch = env
    import zkcensus_amd
    nl, B = 10, 70
    zk, pk, vk = get(nl)
    from census_gen import random_voter
    rng = random.Random(99)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(1, nl), depth_s=rng.randint(1, nl)) for _ in range(B)]
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0] * B
    d_w = dev_bytes(torch, b''.join(ws))
    rs = b''.join(rng.randrange(1 << 248).to_bytes(32, 'little') for _ in range(2 * B))
    ref = pk.prove_batch_dev(d_w.data_ptr(), B, rs)
    old = {k: os.environ.get(k) for k in ('ZKC_LANES', 'ZKC_INFLIGHT')}
    try:
        for lanes, inflight in (('2', '8'), ('1', '3'), ('2', '32'), ('4', '5'), ('3', '96')):
            os.environ['ZKC_LANES'] = lanes; os.environ['ZKC_INFLIGHT'] = inflight
            pk2 = zkcensus_amd.ProvingKey(ctx, zk)
            try:
                assert pk2.prove_batch_dev(d_w.data_ptr(), B, rs) == ref, 'lanes=%s inflight=%s' % (lanes, inflight)
            finally:
                pk2.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def test_fullprove_batch_nl160_three_passes_config3(env):
    """BASELINE configs[2] regime at a size the oracle can follow: zkc_fullprove_batch_dev at nLevels = 160 over B = 200 voters of the
    8 192-voter synthetic census (four pipeline passes of 50 voters, one on each of the key's lanes -- three of 67, 67, 66 until round 5 --: 65 536-bucket H jobs, pass
    boundaries inside the batch).  Every proof goes through the product's batch verifier, the pass-boundary proofs through the oracle's pairing verifier,
    and two of them (first of pass 2, last of the batch) are re-proved by the CPU oracle from the device witness: identical bytes."""
    ctx, get, torch = env
    import zkcensus_amd
    from zkcensus_amd import census, groth16
    nl, B = 160, 200
    zk, pk, vk = get(nl)
    voters = census.synthetic_census(ctx, 8192, nl)[4000:4000 + B]
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = dev_bytes(torch, flat)
    nW = ctx.n_wires(nl)
    d_w = torch.zeros(B * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
    rng = random.Random(1600)
    rs = b''.join(rng.randrange(R).to_bytes(32, 'little') for _ in range(2 * B))
    proofs, pubs = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    assert d_st.cpu().tolist() == [0] * B
    assert groth16.verify_batch(ctx, vk, pubs, proofs)
    per = -(-B // -(-B // pk.pass_size))                                 # proofs per pass (zkc_zkey_pass_info)
    for i in sorted({0, per - 1, per, 2 * per - 1, 2 * per, 66, 67, 133, 134, B - 1}):
        assert ol.verify(vk, pubs[256 * i:256 * i + 256], proofs[256 * i:256 * i + 256]), i
    wt = d_w.view(B, nW * 32)
    dev_w = {i: wt[i].cpu().numpy().tobytes() for i in (per, B - 1)}

    def check(i):
        w = dev_w[i]
        rc, ow = ol.witness(voters[i], nLevels=nl)
        assert rc == 0 and ow == w
        r_i = int.from_bytes(rs[64 * i:64 * i + 32], 'little'); s_i = int.from_bytes(rs[64 * i + 32:64 * i + 64], 'little')
        rc, op, opub = ol.prove(zk, w, r_i, s_i)
        assert rc == 0 and op == proofs[256 * i:256 * i + 256] and opub == pubs[256 * i:256 * i + 256], i
    ol.pmap(check, (per, B - 1))
    # a tampered proof in the middle of the batch is caught by the batch verifier
    bad = bytearray(proofs); bad[256 * 100 + 5] ^= 1
    assert not groth16.verify_batch(ctx, vk, pubs, bytes(bad))


def test_fullprove_batch_nl160_b1024_config2_full_size(env):
    """BASELINE configs[2] at its OWN size and in the bench's form: 1 024 voters of the 8 192-voter synthetic census through zkc_batch_begin /
    zkc_batch_finish, two steps in flight on the two call slots (step 2 is begun before step 1 is finished, with its own witness / status buffers and fresh (r, s)) --
    sixteen passes of 64 voters rotating over the key's four pipeline lanes.  All 2 048 proofs through the product's batch verifier; the first / last proofs and both sides of two pass boundaries of BOTH steps
    byte-equal to the CPU oracle's (its witness from the voter's inputs, its proof from that witness and the step's (r, s)), on host threads."""
    ctx, get, torch = env
    import zkcensus_amd
    from zkcensus_amd import census, groth16
    nl, B = 160, 1024
    zk, pk, vk = get(nl)
    voters = census.synthetic_census(ctx, 8192, nl)[:B]
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = dev_bytes(torch, flat)
    nW = ctx.n_wires(nl)
    d_w = [torch.zeros(B * nW * 32, dtype=torch.uint8, device='cuda') for _ in range(2)]; d_st = [torch.zeros(B, dtype=torch.int32, device='cuda') for _ in range(2)]
    rng = random.Random(1024)
    rs = [b''.join(rng.randrange(R).to_bytes(32, 'little') for _ in range(2 * B)) for _ in range(2)]
    pk.batch_begin(0, d_in.data_ptr(), B, d_w[0].data_ptr(), d_st[0].data_ptr(), rs[0])
    pk.batch_begin(1, d_in.data_ptr(), B, d_w[1].data_ptr(), d_st[1].data_ptr(), rs[1])
    out = [pk.batch_finish(0, B), pk.batch_finish(1, B)]
    per = -(-B // -(-B // pk.pass_size))                                 # 64: the library cuts 1 024 into sixteen equal passes (zkc_zkey_pass_info) rotating over pk.lanes lanes
    picks = (0, per - 1, per, 5 * per - 1, 5 * per, B - 1)
    for k in range(2):
        proofs, pubs = out[k]
        assert d_st[k].cpu().tolist() == [0] * B
        assert groth16.verify_batch(ctx, vk, pubs, proofs), 'step %d' % k
    assert out[0][0] != out[1][0] and out[0][1] == out[1][1]             # fresh (r, s): other proofs, same public signals
    wt = [d_w[k].view(B, nW * 32) for k in range(2)]
    dev_w = {(k, i): wt[k][i].cpu().numpy().tobytes() for k in range(2) for i in picks}

    def check(ki):
        k, i = ki
        rc, ow = ol.witness(voters[i], nLevels=nl)
        assert rc == 0 and ow == dev_w[ki], ki
        r_i = int.from_bytes(rs[k][64 * i:64 * i + 32], 'little'); s_i = int.from_bytes(rs[k][64 * i + 32:64 * i + 64], 'little')
        rc, op, opub = ol.prove(zk, ow, r_i, s_i)
        assert rc == 0 and op == out[k][0][256 * i:256 * i + 256] and opub == out[k][1][256 * i:256 * i + 256], ki
    ol.pmap(check, [(k, i) for k in range(2) for i in picks])
    del d_w, d_in


@pytest.mark.parametrize('B', [1, 2, 97, 193])
def test_fullprove_batch_sizes_around_pass_boundaries_nl10(env, B):
    """Batches that do not fill their passes (1, 2), spill one proof into a second pass (97 -> 49 + 48 after balancing) or into a third (193):
    zkc_fullprove_batch_dev == witness + zkc_prove_batch_dev, every proof accepted by the batch verifier, first and last re-proved by the oracle.
    Also the argument checks of the batch entry points."""
    ctx, get, torch = env
    import zkcensus_amd
    from zkcensus_amd import groth16
    nl = 10
    zk, pk, vk = get(nl)
    from census_gen import random_voter
    rng = random.Random(1000 + B)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(0, nl), depth_s=rng.randint(0, nl)) for _ in range(B)]
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = dev_bytes(torch, flat)
    nW = ctx.n_wires(nl)
    d_w = torch.zeros(B * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
    rs = b''.join(rng.randrange(R).to_bytes(32, 'little') for _ in range(2 * B))
    p1, pub1 = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    assert d_st.cpu().tolist() == [0] * B
    assert groth16.verify_batch(ctx, vk, pub1, p1)
    p2, pub2 = pk.prove_batch_dev(d_w.data_ptr(), B, rs)
    assert p1 == p2 and pub1 == pub2
    wt = d_w.view(B, nW * 32)
    for i in {0, B - 1}:
        w = wt[i].cpu().numpy().tobytes()
        rc, op, opub = ol.prove(zk, w, int.from_bytes(rs[64 * i:64 * i + 32], 'little'), int.from_bytes(rs[64 * i + 32:64 * i + 64], 'little'))
        assert rc == 0 and op == p1[256 * i:256 * i + 256] and opub == pub1[256 * i:256 * i + 256]
    if B == 1:
        lib = pk._lib
        buf = ctypes_buf = __import__('ctypes').create_string_buffer(256)
        assert lib.zkc_prove_batch_dev(pk._h, d_w.data_ptr(), pk.n_vars, 0, rs, buf, None) == 4            # B = 0: ZKC_ERR_BAD_ARG
        assert lib.zkc_prove_batch_dev(pk._h, d_w.data_ptr(), pk.n_vars - 1, 1, rs, buf, None) == 3        # ZKC_ERR_INVALID_WITNESS_LENGTH
        assert lib.zkc_prove_batch_dev(pk._h, d_w.data_ptr(), pk.n_vars, 1, (R).to_bytes(32, 'little') * 2, buf, None) == 4    # r = field order: rejected
        assert lib.zkc_fullprove_batch_dev(pk._h, None, 1, d_w.data_ptr(), d_st.data_ptr(), rs, buf, None) == 4
