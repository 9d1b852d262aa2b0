"""Several devices from one host process (include/zkcensus.h zkc_pool_*, SURVEY.md 8e "one host thread + one HIP stream set per device").
A one-GPU box can still exercise everything but the second card: a pool that lists device 0 twice holds two independent contexts and two resident
keys, splits the batch into the same contiguous blocks and proves them from two host threads at once.  Bytes must equal the single-context path
(and so the oracle's), whichever device proved a voter."""
import json, os, random, sys
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))


@pytest.fixture(scope='module')
def env():
    import torch, zkcensus_amd
    from zkcensus_amd import setup
    _, zp, vp = setup.ensure_test_artifacts(10)
    zk = open(zp, 'rb').read()
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    yield zkcensus_amd, torch, zk, json.load(open(vp)), ctx, pk
    pk.close(); ctx.close()


def voters_of(n, seed, nl=10):
    from census_gen import random_voter
    rng = random.Random(seed)
    return [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(1, nl), depth_s=rng.randint(1, nl)) for _ in range(n)], rng


def single_context(zkcensus_amd, torch, ctx, pk, voters, rs, nl=10):
    import numpy as np
    B = len(voters)
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    d_w = torch.zeros(B * ctx.n_wires(nl) * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
    p, pub = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    return p, pub, d_st.cpu().tolist()


@pytest.mark.parametrize('devices,B', [([0], 5), ([0, 0], 5), ([0, 0], 1), ([0, 0, 0], 200), ([0] * 8, 61)])      # [r4] eight entries: the node of BASELINE configs[3] in one host process (eight contexts, keys and host threads; blocks of 8 and 7 voters)
def test_pool_equals_single_context(env, devices, B):
    zkcensus_amd, torch, zk, vk, ctx, pk = env
    nl = 10
    voters, rng = voters_of(B, 1000 + B + len(devices))
    bad = B // 2 if B >= 5 else None
    if bad is not None:
        voters[bad] = dict(voters[bad]); voters[bad]['availableWeight'] = '0'          # census.circom:72: voteWeight <= availableWeight
    rs = b''.join(rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(2 * B))
    pool = zkcensus_amd.DevicePool(devices, zk)
    try:
        assert pool._lib.zkc_pool_size(pool._h) == len(devices) and pool.n_public == pk.n_public
        p, pub, st = pool.fullprove_batch(voters, rs, nLevels=nl)
        p2, pub2, st2 = pool.fullprove_batch(voters, rs, nLevels=nl)                   # buffers are reused: same bytes again
    finally:
        pool.close()
    q, qpub, qst = single_context(zkcensus_amd, torch, ctx, pk, voters, rs)
    assert st == qst == st2 and (bad is None or st[bad] != 0) and all(x == 0 for i, x in enumerate(st) if i != bad)
    good = [i for i in range(B) if i != bad]
    W = 32 * pk.n_public
    assert all(p[256 * i:256 * i + 256] == q[256 * i:256 * i + 256] == p2[256 * i:256 * i + 256] and pub[W * i:W * i + W] == qpub[W * i:W * i + W] for i in good)
    for i in sorted({good[0], good[-1], good[len(good) // 3]}):                       # one voter of each device's block
        assert ol.verify(vk, pub[W * i:W * i + W], p[256 * i:256 * i + 256])


def test_pool_draws_uniform_blinding_when_none_is_given(env):
    zkcensus_amd, torch, zk, vk, ctx, pk = env
    voters, _ = voters_of(3, 5)
    pool = zkcensus_amd.DevicePool([0, 0], zk)
    try:
        a, pub, st = pool.fullprove_batch(voters, None, nLevels=10)
        b, _, _ = pool.fullprove_batch(voters, None, nLevels=10)
    finally:
        pool.close()
    W = 32 * pk.n_public
    assert st == [0, 0, 0] and a != b                                                   # fresh r, s per call
    assert all(ol.verify(vk, pub[W * i:W * i + W], x[256 * i:256 * i + 256]) for x in (a, b) for i in range(3))


def test_pool_errors(env):
    zkcensus_amd, torch, zk, vk, ctx, pk = env
    with pytest.raises(zkcensus_amd.ZkcError):
        zkcensus_amd.DevicePool([99])                                                   # no such device
    pool = zkcensus_amd.DevicePool([0])
    try:
        voters, _ = voters_of(1, 6)
        pool.n_public = 8
        with pytest.raises(zkcensus_amd.ZkcError, match='no key loaded'):
            pool.fullprove_batch(voters, None, nLevels=10)
        with pytest.raises(zkcensus_amd.ZkcError):
            pool.load_key(zk[:4096])                                                    # truncated image: every device or none
        assert not pool._lib.zkc_pool_zkey(pool._h, 0)
        pool.load_key(zk)
        p, pub, st = pool.fullprove_batch(voters, None, nLevels=10)
        assert st == [0] and ol.verify(vk, pub, p)
    finally:
        pool.close()
