"""GPU: batch verifier -- N proofs folded into one pairing-product check.  The oracle verifies each proof
singly (pinned pairing verifier); the batch result must be the AND of those verdicts, for honest batches and for batches with one
tampered member (proof point swapped for another valid curve point, public signal changed, point off the curve)."""
import json, os, random, sys
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))


def test_batch_verify_matches_single_verdicts():
    import torch, numpy as np  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import groth16, setup
    from census_gen import random_voter
    nl, N = 10, 24
    ctx = zkcensus_amd.Context(0)
    _, zp, vp = setup.ensure_test_artifacts(nl)
    zk = open(zp, 'rb').read(); pk = zkcensus_amd.ProvingKey(ctx, zk); vk = json.load(open(vp))
    rng = random.Random(42)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(1, nl), depth_s=rng.randint(1, nl)) for _ in range(N)]
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0] * N
    d = torch.from_numpy(np.frombuffer(b''.join(ws), dtype=np.uint8).copy()).cuda()
    rs = b''.join(rng.randrange(1 << 248).to_bytes(32, 'little') for _ in range(2 * N))
    proofs, pubs = pk.prove_batch_dev(d.data_ptr(), N, rs)
    npub = 8
    single = [ol.verify(vk, pubs[32 * npub * i:32 * npub * (i + 1)], proofs[256 * i:256 * (i + 1)]) for i in range(N)]
    assert all(single)
    seed = bytes(range(32))
    for path in ('0', '1'):            # the Miller loops on host threads, then on the GPU (zkc_pairing_dev.hip; by default from 128 proofs on): the same verdict on every case below
        os.environ['ZKC_VERIFY_BATCH_GPU'] = path
        try:
            _verdicts(ctx, groth16, vk, pubs, proofs, seed, npub)
        finally:
            del os.environ['ZKC_VERIFY_BATCH_GPU']
    pk.close(); ctx.close()


def _verdicts(ctx, groth16, vk, pubs, proofs, seed, npub):
    assert groth16.verify_batch(ctx, vk, pubs, proofs, seed) is True
    assert groth16.verify_batch(ctx, vk, pubs, proofs) is True                       # weights from the OS
    assert groth16.verify_batch(ctx, vk, pubs[:32 * npub], proofs[:256], seed) is True   # N = 1

    def with_patch(buf, off, new):
        b = bytearray(buf); b[off:off + len(new)] = new; return bytes(b)
    # (1) proof 5 gets proof 7's C: every point is still on the curve, only the pairing equation can tell
    bad = with_patch(proofs, 256 * 5 + 192, proofs[256 * 7 + 192:256 * 7 + 256])
    assert not ol.verify(vk, pubs[32 * npub * 5:32 * npub * 6], bad[256 * 5:256 * 6])
    assert groth16.verify_batch(ctx, vk, pubs, bad, seed) is False
    # (2) proofs 2 and 3 exchange their A points
    a2, a3 = proofs[256 * 2:256 * 2 + 64], proofs[256 * 3:256 * 3 + 64]
    bad = with_patch(with_patch(proofs, 256 * 2, a3), 256 * 3, a2)
    assert groth16.verify_batch(ctx, vk, pubs, bad, seed) is False
    # (3) one bit of one public signal
    off = 32 * npub * 11 + 32 * 2
    badpub = with_patch(pubs, off, bytes([pubs[off] ^ 1]))
    assert not ol.verify(vk, badpub[32 * npub * 11:32 * npub * 12], proofs[256 * 11:256 * 12])
    assert groth16.verify_batch(ctx, vk, badpub, proofs, seed) is False
    # (4) a point off the curve, a coordinate >= q, a public signal >= r: rejected before any pairing
    assert groth16.verify_batch(ctx, vk, pubs, with_patch(proofs, 256 * 9, bytes([proofs[256 * 9] ^ 1])), seed) is False
    assert groth16.verify_batch(ctx, vk, pubs, with_patch(proofs, 256 * 9, b'\xff' * 32), seed) is False
    assert groth16.verify_batch(ctx, vk, with_patch(pubs, 0, b'\xff' * 32), proofs, seed) is False
    # (5) the negation of a valid A (still in G1): e(-A, B) flips
    q = ol.Q
    y = int.from_bytes(proofs[32:64], 'little')
    assert groth16.verify_batch(ctx, vk, pubs, with_patch(proofs, 32, ((q - y) % q).to_bytes(32, 'little')), seed) is False
    # (6) a B on the twist but outside G2 (the twist has a cofactor): refused by the membership test, as zkc_verify_bin refuses it
    pt = ol.twist_point_outside_g2()
    offg2 = b''.join(ol.le32(v) for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]))
    assert groth16.verify_batch(ctx, vk, pubs, with_patch(proofs, 256 * 4 + 64, offg2), seed) is False
    # (7) a B at infinity contributes 1 to the product: the batch fails on that proof's equation, not on a crash
    assert groth16.verify_batch(ctx, vk, pubs, with_patch(proofs, 256 * 6 + 64, bytes(128)), seed) is False


def test_batch_verify_gpu_path_odd_sizes_and_several_rounds():
    """[r5] The default path from 128 proofs on (Miller loops on the GPU, csrc/zkc_pairing_dev.hip) at sizes that are not powers of two -- odd levels in the product tree --
    and with the pairs taken in several rounds of kernels (ZKC_VERIFY_CHUNK shrinks the 16 384 of a round): the verdicts of the host-thread path, which the test above pins to the
    oracle's, on an honest batch and on batches with one bad member at the start, in the middle (across a round boundary) and at the end."""
    import torch, numpy as np  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import groth16, setup
    from census_gen import random_voter
    nl, base = 10, 12
    ctx = zkcensus_amd.Context(0)
    _, zp, vp = setup.ensure_test_artifacts(nl)
    pk = zkcensus_amd.ProvingKey(ctx, open(zp, 'rb').read()); vk = json.load(open(vp))
    rng = random.Random(7)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randint(1, nl), depth_s=rng.randint(1, nl)) for _ in range(base)]
    ws, st = ctx.witness(voters, nLevels=nl); assert st == [0] * base
    d = torch.from_numpy(np.frombuffer(b''.join(ws), dtype=np.uint8).copy()).cuda()
    rs = b''.join(rng.randrange(1 << 248).to_bytes(32, 'little') for _ in range(2 * base))
    proofs, pubs = pk.prove_batch_dev(d.data_ptr(), base, rs)
    assert all(ol.verify(vk, pubs[256 * i:256 * (i + 1)], proofs[256 * i:256 * (i + 1)]) for i in range(base))
    for N, chunk in ((129, None), (301, '100'), (257, '2')):
        P = b''.join(proofs[256 * (i % base):256 * (i % base + 1)] for i in range(N)); U = b''.join(pubs[256 * (i % base):256 * (i % base + 1)] for i in range(N))
        if chunk:
            os.environ['ZKC_VERIFY_CHUNK'] = chunk
        try:
            assert groth16.verify_batch(ctx, vk, U, P) is True
            for bad_at in (0, 100, N - 1):
                other = (bad_at + 1) % base
                Pb = bytearray(P); Pb[256 * bad_at + 192:256 * bad_at + 256] = proofs[256 * other + 192:256 * other + 256]        # another proof's C
                if bad_at % base == other:
                    continue
                assert groth16.verify_batch(ctx, vk, U, bytes(Pb)) is False, (N, chunk, bad_at)
                os.environ['ZKC_VERIFY_BATCH_GPU'] = '0'
                try:
                    assert groth16.verify_batch(ctx, vk, U, bytes(Pb)) is False                      # the host-thread path agrees
                finally:
                    del os.environ['ZKC_VERIFY_BATCH_GPU']
        finally:
            os.environ.pop('ZKC_VERIFY_CHUNK', None)
    pk.close(); ctx.close()
