"""The prover is not specific to the census circuit: groth16.prove(zkey, wtns) (ts_inputs/src/example.ts:358 via fullProve; rapidsnark's groth16_prover) takes ANY Groth16 key.
Random satisfiable R1CS instances of several sizes -- down to a domain of 8 points -- go through the test-only setup, the C oracle, the toxic-waste closed form and (on a GPU) the
product's unfolded path: identical proof bytes, accepted by the pinned verifier."""
import ctypes, json, os, random
import pytest
import oracle_lib as ol
import closed_form as cf


def random_instance(tmp_path, n_cons, n_wires, n_pub, seed):
    """-> (.r1cs path, witness bytes): every constraint <A,w> <B,w> = <C,w> with sparse random A, B and a C row of one or two wires solved for the product"""
    from zkcensus_amd import r1cs
    rng = random.Random(seed)
    w = [1] + [rng.randrange(1, ol.R) for _ in range(n_wires - 1)]
    cs = r1cs.R1CS(n_wires, n_pub)
    for k in range(n_cons):
        a = {rng.randrange(n_wires): rng.randrange(1, ol.R) for _ in range(rng.randrange(1, 5))}
        b = {rng.randrange(n_wires): rng.randrange(1, ol.R) for _ in range(rng.randrange(1, 4))}
        if k % 7 == 0:
            a = {0: rng.randrange(1, 50)}                                       # constants on the left: rows with the "one" wire
        av = sum(c * w[i] for i, c in a.items()) % ol.R; bv = sum(c * w[i] for i, c in b.items()) % ol.R
        i1, i2 = rng.randrange(1, n_wires), rng.randrange(n_wires)
        c2 = rng.randrange(ol.R) if i2 != i1 else 0
        c1 = (av * bv - c2 * w[i2]) * pow(w[i1], ol.R - 2, ol.R) % ol.R
        c = {i1: c1}
        if c2:
            c[i2] = c2
        cs.add(a, b, c)
    assert cs.check(w) == -1
    path = str(tmp_path / ('generic_%d_%d.r1cs' % (n_cons, n_wires)))
    cs.write(path)
    return path, b''.join(x.to_bytes(32, 'little') for x in w)


def setup_key(r1cs_path, seed):
    from zkcensus_amd import _native
    z, v = r1cs_path[:-5] + '.zkey', r1cs_path[:-5] + '_vkey.json'
    err = ctypes.create_string_buffer(512)
    rc = _native.load().zkc_setup_from_r1cs(r1cs_path.encode(), seed, z.encode(), v.encode(), err, 512)
    assert rc == 0, err.value
    return open(z, 'rb').read(), json.load(open(v))


SIZES = [(3, 6, 1), (5, 9, 2), (50, 64, 3), (40, 1500, 2), (300, 200, 0), (1000, 900, 5), (5000, 3000, 8), (12000, 9000, 2)]       # domains 8 .. 2^14; (40, 1500, 2): far more wires than 3 x the domain (most of them in no constraint: bases at infinity)
GPU_SIZES = SIZES + [(28000, 20000, 1), (50000, 40000, 6)]                                                                  # .. 2^15, 2^16 (2^18 and up: tests/test_gpu_generic_large.py)


@pytest.mark.parametrize('n_cons,n_wires,n_pub', SIZES)
def test_oracle_and_closed_form_on_random_circuits(tmp_path, n_cons, n_wires, n_pub):
    r1, w = random_instance(tmp_path, n_cons, n_wires, n_pub, seed=n_cons)
    zk, vk = setup_key(r1, 4242 + n_cons)
    for r, s in ((1, 2), (ol.R - 3, 12345678901234567890)):
        rc, proof, pub = ol.prove(zk, w, r, s, npub=max(n_pub, 1)); assert rc == 0
        pub = pub[:32 * n_pub]
        a, b, c = cf.proof_scalars(r1, 4242 + n_cons, w, r, s)
        assert proof == cf.proof_from_scalars(ol, a, b, c)
        assert ol.verify(vk, pub, proof) if n_pub else True


@pytest.mark.gpu
@pytest.mark.parametrize('n_cons,n_wires,n_pub', GPU_SIZES)
def test_gpu_prover_on_random_circuits(tmp_path, n_cons, n_wires, n_pub):
    import torch  # noqa: F401
    import zkcensus_amd
    r1, w = random_instance(tmp_path, n_cons, n_wires, n_pub, seed=n_cons)
    zk, vk = setup_key(r1, 4242 + n_cons)
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    assert pk.n_vars == n_wires and pk.n_public == n_pub
    for r, s in ((1, 2), (ol.R - 3, 12345678901234567890)):
        proof, pub = pk.prove(w, r, s)
        a, b, c = cf.proof_scalars(r1, 4242 + n_cons, w, r, s)
        assert proof == cf.proof_from_scalars(ol, a, b, c), 'GPU proof of a generic circuit differs from the closed form'
        assert pub == w[32:32 * (1 + n_pub)]
    # a small batch through the same key, bytes per proof equal the single calls
    import numpy as np
    B = 5
    d_w = torch.from_numpy(np.frombuffer(w * B, dtype=np.uint8).copy()).cuda()
    rs = b''.join(int(x).to_bytes(32, 'little') for k in range(B) for x in (7 + k, 9 + k))
    proofs, pubs = pk.prove_batch_dev(d_w.data_ptr(), B, rs)
    for k in range(B):
        p1, _ = pk.prove(w, 7 + k, 9 + k)
        assert proofs[256 * k:256 * k + 256] == p1
    pk.close(); ctx.close()
