"""GPU parity: the HIP witness kernels (through the C ABI) vs the CPU oracle and the reference-wasm golden vectors."""
import hashlib, random
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
VEC = ol.load_json('witness_vectors.json')


@pytest.fixture(scope='module')
def ctx():
    import zkcensus_amd
    c = zkcensus_amd.Context(0)
    yield c
    c.close()


def test_shape(ctx):
    assert ctx.n_wires(160) == 82754 and ctx.n_inputs(160) == 334
    assert ctx.n_wires(160) == ol.lib().zko_n_wires(160)
    for nl in (10, 32, 100, 252, 253):
        assert ctx.n_wires(nl) == ol.lib().zko_n_wires(nl)


def test_golden_vectors_batch(ctx):
    vecs = VEC['vectors']
    ws, st = ctx.witness([v['inputs'] for v in vecs])
    assert st == [0] * len(vecs)
    for v, w in zip(vecs, ws):
        assert hashlib.sha256(w).hexdigest() == v['sha256'], v['name']
        assert [str(int.from_bytes(w[32 * i:32 * i + 32], 'little')) for i in range(1, 9)] == v['public']


def test_negative_vectors(ctx):
    """every single, pair and triple of violated asserts: the status names the assert the reference's wasm reaches first (its own message is in the fixture)"""
    negs = VEC['negative']
    good = VEC['vectors'][0]['inputs']
    ws, st = ctx.witness([good] + [v['inputs'] for v in negs] + [good])
    assert st[0] == 0 and st[-1] == 0                # failures do not poison the batch
    assert st[1:-1] == [ol.status_of_wasm_message(v['wasm_msg']) for v in negs]
    assert st[1:7] == [1, 3, 2, 4, 5, 7]
    assert st[1:-1] == [ol.witness(v['inputs'])[0] for v in negs]
    assert hashlib.sha256(ws[0]).hexdigest() == VEC['vectors'][0]['sha256'] == hashlib.sha256(ws[-1]).hexdigest()


def test_random_voters_vs_oracle(ctx):
    import sys, os
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    rng = random.Random(20261003)
    voters = [random_voter(rng, ol.poseidon, depth_c=rng.randrange(0, 40), depth_s=rng.randrange(0, 40)) for _ in range(96)]
    voters += [random_voter(rng, ol.poseidon, depth_c=160, depth_s=160, zero_frac=0.0)]
    ws, st = ctx.witness(voters)
    assert st == [0] * len(voters)
    for v, w in zip(voters, ws):
        rc, wo = ol.witness(v)
        assert rc == 0 and w == wo


def test_other_depths_vs_oracle(ctx):
    import sys, os
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    from census_gen import random_voter
    rng = random.Random(7)
    for nl in (10, 31, 253):                               # 253: the largest circuit circomlib permits (key bit 253 is the solved bit there, not a wire)
        voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(0, nl + 1), depth_s=rng.randrange(0, nl + 1)) for _ in range(8 if nl < 100 else 3)]
        if nl == 253:
            voters.append(random_voter(rng, ol.poseidon, nLevels=nl, depth_c=253, depth_s=253))
        ws, st = ctx.witness(voters, nLevels=nl)
        assert st == [0] * len(voters)
        for v, w in zip(voters, ws):
            rc, wo = ol.witness(v, nLevels=nl)
            assert rc == 0 and w == wo


def test_lane_and_wave_kernels_agree(ctx):
    """Two witness kernels share the C ABI entry: a wave per Merkle path for launches of up to 128 voters (latency), a lane per path above
    that (throughput).  Same voters through both -- including the reference-wasm vectors and the rejected inputs -- must give the same bytes
    and the same statuses."""
    import numpy as np, torch, zkcensus_amd
    vecs = [v['inputs'] for v in VEC['vectors']] + [v['inputs'] for v in VEC['negative']]
    many = (vecs * 7)[:150]
    ws_wave, st_wave = ctx.witness(many)                      # stand-alone call, 150 <= 1024: wave-per-path kernel
    # the lane-per-path kernel is what the batch pipeline uses for launches of more than 128 voters: reach it through zkc_fullprove_batch_dev's
    # witness stage (first pass: wave; the rest of the group: lane) with a batch of 300 = passes of 75
    from zkcensus_amd import setup
    _, zp, _ = setup.ensure_test_artifacts(160)
    pk = zkcensus_amd.ProvingKey(ctx, open(zp, 'rb').read())
    big = (many * 2)[:300]
    flat = b''.join(zkcensus_amd.flatten_inputs(v) for v in big)
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    nW = ctx.n_wires(160)
    d_w = torch.zeros(300 * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(300, dtype=torch.int32, device='cuda')
    pk.fullprove_batch_dev(d_in.data_ptr(), 300, d_w.data_ptr(), d_st.data_ptr(), bytes(64 * 300))
    pk.close()
    st_lane = d_st.cpu().tolist()[150:300]                    # voters 150..299 = `many` again, produced by the lane kernel (passes 2, 3 of the group)
    wl = d_w.view(300, nW * 32)[150:300].cpu().numpy()
    ws_lane = [wl[i].tobytes() for i in range(150)]
    assert st_lane == st_wave
    nv = len(VEC['vectors'])
    for i, (a, b) in enumerate(zip(ws_lane, ws_wave)):
        if st_lane[i] == 0:
            assert a == b, i
            assert hashlib.sha256(a).hexdigest() == VEC['vectors'][i % len(vecs)]['sha256']
    assert sum(1 for s in st_lane if s) == len([i for i in range(150) if (i % len(vecs)) >= nv])
