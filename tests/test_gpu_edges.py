"""GPU: the edges of the boundary that the main parity tests do not visit -- empty and malformed calls, input values at and above the field order, the
shallowest and deepest voters, the extreme blinding scalars, a witness of the wrong length.  Everything goes through the C ABI (ctypes) and is compared
with the CPU oracle where there is something to compare."""
import ctypes, json, os, random, sys
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))


@pytest.fixture(scope='module')
def env():
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import setup
    nl = 10
    _, zp, vp = setup.ensure_test_artifacts(nl)
    zk = open(zp, 'rb').read()
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    yield zkcensus_amd, ctx, pk, zk, json.load(open(vp)), nl
    pk.close(); ctx.close()


def test_empty_and_null_calls_are_refused_not_crashed(env):
    zkc, ctx, pk, zk, vk, nl = env
    lib = ctx._lib
    nw, ni = ctx.n_wires(nl), ctx.n_inputs(nl)
    buf = ctypes.create_string_buffer(nw * 32); st = (ctypes.c_int32 * 1)()
    assert lib.zkc_witness(ctx._h, nl, bytes(ni * 32), 0, buf, st) == 4                       # B = 0: ZKC_ERR_BAD_ARG
    assert lib.zkc_witness(ctx._h, nl, None, 1, buf, st) == 4
    assert lib.zkc_witness(ctx._h, 2, bytes(ni * 32), 1, buf, st) == 4                         # nLevels below the circuit's minimum
    assert lib.zkc_witness(ctx._h, 254, bytes(ni * 32), 1, buf, st) == 4                       # above the largest circuit circomlib permits (253)
    proof = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(256)
    assert lib.zkc_prove_batch_dev(pk._h, None, nw, 1, bytes(64), proof, pub) == 4
    assert lib.zkc_prove_batch_dev(pk._h, 1, nw, 0, bytes(64), proof, pub) == 4                # B = 0
    assert lib.zkc_zkey_load(ctx._h, b'', 0, ctypes.byref(ctypes.c_void_p())) in (4, 5)
    assert lib.zkc_zkey_load(ctx._h, zk[:11], 11, ctypes.byref(ctypes.c_void_p())) == 5       # ZKC_ERR_FORMAT
    # and the context still works
    from census_gen import random_voter
    ws, s = ctx.witness([random_voter(random.Random(1), ol.poseidon, nLevels=nl, depth_c=3, depth_s=3)], nLevels=nl)
    assert s == [0]


def test_input_values_at_and_above_the_field_order(env):
    """census.circom's inputs are field elements: the C ABI takes 32-byte values below r.  r - 1 is a legal value (and fails the circuit's own checks, not the range check);
    r and 2^256 - 1 are refused with ZKC_W_ERR_INPUT_RANGE for that voter alone, exactly where the oracle refuses them."""
    zkc, ctx, pk, zk, vk, nl = env
    from census_gen import random_voter
    good = zkc.flatten_inputs(random_voter(random.Random(2), ol.poseidon, nLevels=nl, depth_c=4, depth_s=2), nl)
    ni = ctx.n_inputs(nl)

    def patched(i, value):
        b = bytearray(good); b[32 * i:32 * i + 32] = value.to_bytes(32, 'little'); return bytes(b)
    cases = [good, patched(4, ol.R), patched(4, 2**256 - 1), patched(4, ol.R - 1), patched(ni - 1, ol.R), good]      # voteHash[0] (unconstrained), last sikSibling
    ws, st = ctx.witness(cases, nLevels=nl)
    ost = [ol.witness(c, nl)[0] for c in cases]
    assert st == ost == [0, 6, 6, 0, 6, 0]
    assert ws[0] == ws[5] == ol.witness(good, nl)[1] and ws[3] == ol.witness(cases[3], nl)[1]


def test_shallowest_and_deepest_voters(env):
    """depth 0 (a census of one voter: every sibling zero, the root IS the leaf hash) and depth nLevels in both trees; witness and proof bytes equal the oracle's"""
    zkc, ctx, pk, zk, vk, nl = env
    from census_gen import random_voter
    rng = random.Random(3)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=0, depth_s=0), random_voter(rng, ol.poseidon, nLevels=nl, depth_c=nl, depth_s=nl),
              random_voter(rng, ol.poseidon, nLevels=nl, depth_c=0, depth_s=nl), random_voter(rng, ol.poseidon, nLevels=nl, depth_c=1, depth_s=1, zero_frac=0.0)]
    assert all(s == '0' for s in voters[0]['censusSiblings'])
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0] * 4
    for v, w in zip(voters, ws):
        rc, ow = ol.witness(v, nl); assert rc == 0 and ow == w
        p, u = pk.prove(w, 7, 9)
        rc, op, ou = ol.prove(zk, w, 7, 9); assert rc == 0 and (p, u) == (op, ou)
        assert ol.verify(vk, u, p)


def test_extreme_blinding_scalars_and_bad_ones(env):
    zkc, ctx, pk, zk, vk, nl = env
    from census_gen import random_voter
    v = random_voter(random.Random(4), ol.poseidon, nLevels=nl, depth_c=5, depth_s=6)
    ws, st = ctx.witness([v], nLevels=nl)
    for r, s in ((0, 0), (0, 1), (ol.R - 1, ol.R - 1), (1, ol.R - 1), (2**253, 2**128)):
        p, u = pk.prove(ws[0], r, s)
        rc, op, ou = ol.prove(zk, ws[0], r, s)
        assert rc == 0 and (p, u) == (op, ou), (r, s)
        assert ol.verify(vk, u, p)
    with pytest.raises(zkc.ZkcError) as e:                                                    # r = the field order itself: refused, not reduced silently
        pk.prove(ws[0], ol.R, 1)
    assert e.value.code == 4
    with pytest.raises(zkc.ZkcError) as e:
        pk.prove(ws[0][:-32], 1, 2)                                                            # one wire short: rapidsnark's INVALID_WITNESS_LENGTH
    assert e.value.code == 3 and 'Invalid witness length' in str(e.value)


def test_ragged_sibling_lists_flatten_like_the_reference_pads_them(env):
    """internal/inputs.go:52,72 appends zero siblings up to nLevels + 1; a caller may hand over the packed list.  Short, exact and over-long lists."""
    zkc, ctx, pk, zk, vk, nl = env
    from census_gen import random_voter
    v = random_voter(random.Random(5), ol.poseidon, nLevels=nl, depth_c=3, depth_s=2)
    packed = dict(v, censusSiblings=v['censusSiblings'][:3], sikSiblings=v['sikSiblings'][:2])
    assert zkc.flatten_inputs(packed, nl) == zkc.flatten_inputs(v, nl) == ol.flat_inputs(packed, nl)
    ws, st = ctx.witness([packed, v], nLevels=nl)
    assert st == [0, 0] and ws[0] == ws[1]
    with pytest.raises((ValueError, AssertionError)):
        zkc.flatten_inputs(dict(v, censusSiblings=v['censusSiblings'] + ['0']), nl)


def test_split_batch_calls_overlap_and_match_the_synchronous_call(env):
    """zkc_batch_begin / zkc_batch_finish: two calls in flight on one key (two slots, two sets of buffers) give the bytes of two synchronous calls; a slot that is busy
    cannot be begun again and an idle one cannot be finished."""
    zkc, ctx, pk, zk, vk, nl = env
    import torch, numpy as np
    from census_gen import random_voter
    rng = random.Random(6)
    nW = ctx.n_wires(nl)
    batches = []
    for B in (5, 9):
        voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(0, nl + 1), depth_s=rng.randrange(0, nl + 1)) for _ in range(B)]
        flat = b''.join(zkc.flatten_inputs(v, nl) for v in voters)
        rs = b''.join(rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(2 * B))
        d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
        d_w = torch.empty(B * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
        batches.append((B, voters, rs, d_in, d_w, d_st))
    sync = [pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs) for B, _, rs, d_in, d_w, d_st in batches]
    for (B, _, rs, d_in, d_w, d_st), slot in zip(batches, (0, 1)):
        pk.batch_begin(slot, d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    with pytest.raises(zkc.ZkcError) as e:                              # slot 1 has a call in flight
        B, _, rs, d_in, d_w, d_st = batches[0]
        pk.batch_begin(1, d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    assert e.value.code == 4
    got = [pk.batch_finish(0, batches[0][0]), pk.batch_finish(1, batches[1][0])]
    assert got == sync
    with pytest.raises(zkc.ZkcError):
        pk.batch_finish(0, 5)                                           # nothing in flight on slot 0 any more
    # [r4] a call of several proofs with a call of ONE proof begun right behind it, forty times: the lone call blinds on the G1 stream, the larger one on the blinding stream,
    # and both use the lane's one blinding scratch -- the lone call has to wait for the other's blinding (round 3 did not: a wrong proof once in a few hundred service batches)
    B9, _, rs9, d_in9, d_w9, d_st9 = batches[1]
    d_w1 = torch.empty(nW * 32, dtype=torch.uint8, device='cuda'); d_st1 = torch.zeros(1, dtype=torch.int32, device='cuda')
    lone_sync = pk.fullprove_batch_dev(d_in9.data_ptr(), 1, d_w1.data_ptr(), d_st1.data_ptr(), rs9[:64])
    assert lone_sync[0] == sync[1][0][:256]
    for k in range(40):
        pk.batch_begin(k & 1, d_in9.data_ptr(), B9, d_w9.data_ptr(), d_st9.data_ptr(), rs9)
        pk.batch_begin(1 - (k & 1), d_in9.data_ptr(), 1, d_w1.data_ptr(), d_st1.data_ptr(), rs9[:64])
        assert pk.batch_finish(k & 1, B9) == sync[1], k
        assert pk.batch_finish(1 - (k & 1), 1) == lone_sync, k
    # witnesses given (d_inputs = NULL): the groth16.prove shape through the same two halves
    B, voters, rs, d_in, d_w, d_st = batches[1]
    pk.batch_begin(0, None, B, d_w.data_ptr(), None, rs)
    assert pk.batch_finish(0, B) == sync[1]
    rc, w = ol.witness(voters[3], nl)
    rc2, op, ou = ol.prove(zk, w, int.from_bytes(rs[64 * 3:64 * 3 + 32], 'little'), int.from_bytes(rs[64 * 3 + 32:64 * 4], 'little'))
    assert rc == 0 and rc2 == 0 and sync[1][0][256 * 3:256 * 4] == op


def test_small_pass_blinding_without_variable_base_products_matches_the_general_one(env):
    """Passes of one or two proofs take their own blinding (zkc_finalize.hip: two more MSM jobs, 4-bit fixed-base tables summed by a butterfly over the lanes, division on the
    host); ZKC_BLIND_TREE=0 at key load selects the general kernels.  Same bytes for one and two proofs, for this
    build's witnesses (folded) and a foreign one (not folded), for (r, s) = (0, 0), small and full-size, and for the stub pass at the end of a larger call."""
    zkc, ctx, pk, zk, vk, nl = env
    import torch, numpy as np
    from census_gen import random_voter
    rng = random.Random(8)
    nW = ctx.n_wires(nl)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(0, nl + 1), depth_s=rng.randrange(0, nl + 1)) for _ in range(5)]
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0] * 5
    foreign = b''.join([(1).to_bytes(32, 'little')] + [rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(nW - 1)])      # not a witness of the circuit: proved all the same, nothing folded
    cases = [([ws[0]], [(0, 0)]), ([ws[0], ws[1]], [(1, 2), (ol.R - 1, ol.R - 2)]), ([bytes(foreign)], [(rng.randrange(ol.R), rng.randrange(ol.R))]),
             ([bytes(foreign), ws[2]], [(5, 0), (0, 7)]), (list(ws), [(rng.randrange(ol.R), rng.randrange(ol.R)) for _ in range(5)])]
    old = {k: os.environ.get(k) for k in ('ZKC_BLIND_TREE', 'ZKC_INFLIGHT')}
    try:
        os.environ['ZKC_BLIND_TREE'] = '0'
        pk_general = zkc.ProvingKey(ctx, zk)
        os.environ.pop('ZKC_BLIND_TREE'); os.environ['ZKC_INFLIGHT'] = '2'          # five proofs = passes of 2, 2 and a stub of 1, every one of them a small pass
        pk_stub = zkc.ProvingKey(ctx, zk)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    try:
        for wl, rsl in cases:
            rs = b''.join(r.to_bytes(32, 'little') + s.to_bytes(32, 'little') for r, s in rsl)
            d_w = torch.from_numpy(np.frombuffer(b''.join(wl), dtype=np.uint8).copy()).cuda()
            got = pk.prove_batch_dev(d_w.data_ptr(), len(wl), rs)
            assert got == pk_general.prove_batch_dev(d_w.data_ptr(), len(wl), rs) == pk_stub.prove_batch_dev(d_w.data_ptr(), len(wl), rs), (len(wl), rsl)
        rc, op, ou = ol.prove(zk, ws[0], 0, 0)
        d_w = torch.from_numpy(np.frombuffer(ws[0], dtype=np.uint8).copy()).cuda()
        assert rc == 0 and pk.prove_batch_dev(d_w.data_ptr(), 1, bytes(64))[0] == op
    finally:
        pk_general.close(); pk_stub.close()


def test_lone_calls_soak_against_the_oracle(env):
    """Sixty inputs -> proof calls of one or two voters (the path that lays its pass out from the inputs' sibling depths while the witness kernel runs, blinds without variable-base
    products, takes the 8-bit-window G2 table and lets the host divide): random depths 0..nLevels in both trees, random (r, s), every sixth voter rejected by the circuit (weight),
    one call with a non-zero LAST sibling (no early layout for that call).  Every accepted voter's proof equals the CPU oracle's bytes; a rejected voter fails alone."""
    zkc, ctx, pk, zk, vk, nl = env
    import torch, numpy as np
    from census_gen import random_voter
    rng = random.Random(20261004)
    nW, nIn = ctx.n_wires(nl), ctx.n_inputs(nl)
    d_w = torch.empty(2 * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(2, dtype=torch.int32, device='cuda')
    k = 0; todo = []
    for call in range(60):
        B = 1 + (call & 1)
        voters = []
        for _ in range(B):
            v = random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(0, nl + 1), depth_s=rng.randrange(0, nl + 1))
            if k % 6 == 5: v = dict(v, voteWeight=str(int(v['availableWeight']) + 1))
            if call == 31: v = dict(v, sikSiblings=list(v['sikSiblings'][:nl]) + ['9'])
            voters.append(v); k += 1
        flat = b''.join(zkc.flatten_inputs(v, nl) for v in voters)
        rs = [(rng.randrange(ol.R), rng.randrange(ol.R)) for _ in range(B)]
        d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
        proofs, pubs = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), b''.join(r.to_bytes(32, 'little') + s.to_bytes(32, 'little') for r, s in rs))
        st = d_st.cpu().tolist()[:B]
        for q, v in enumerate(voters):
            todo.append((call, q, v, st[q], rs[q], proofs[256 * q:256 * q + 256], pubs[256 * q:256 * q + 256]))

    def check(rec):                                   # the oracle's half, on host threads after the GPU's sixty calls
        call, q, v, st, (r, s), proof, pub = rec
        rc, w = ol.witness(v, nl)
        assert st == rc, (call, q)
        if rc == 0:
            rc2, op, ou = ol.prove(zk, w, r, s)
            assert rc2 == 0 and proof == op and pub == ou, (call, q)
    ol.pmap(check, todo)
