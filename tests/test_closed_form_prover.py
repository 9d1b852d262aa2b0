"""The independent prover oracle SURVEY.md section 7 step 2 / BASELINE.md section 5 promise: GPU proof == own CPU restatement == toxic-waste closed form.

tests/closed_form.py evaluates the three proof elements as SCALARS from (seed, .r1cs, witness, r, s) with Lagrange evaluation at tau -- no transform, no
multi-scalar multiplication, no .zkey -- and one scalar multiplication of the generator each turns them into the 256 proof bytes.  The CPU test pins
the C oracle's prove stage that way (so it is no longer checked by self-verification alone); the GPU tests pin the product's at nLevels 10 and 160."""
import json, os, random, sys
import pytest
import oracle_lib as ol
import closed_form as cf

sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))


def _artifacts(nl):
    from zkcensus_amd import setup
    return setup.ensure_test_artifacts(nl), setup.DEFAULT_SEED


def _voter(nl, seed, **kw):
    from census_gen import random_voter
    return random_voter(random.Random(seed), ol.poseidon, nLevels=nl, **kw)


def test_setup_key_is_the_seed_s_key():
    """alpha, beta, delta of the generated key are the closed form's: vk_alpha_1 = alpha G1, vk_delta_2 = delta G2 (so the seed really is the key's toxic waste)"""
    (r1, zk, vkp), seed = _artifacts(10)
    vk = json.load(open(vkp))
    tau, alpha, beta, gamma, delta = cf.toxic_waste(seed)
    assert ol.g1_mul(cf.G1_GEN, alpha) == ol.g1_json(vk['vk_alpha_1'])
    assert ol.msm_g2(cf.G2_GEN, ol.le32(beta)) == ol.g2_json(vk['vk_beta_2'])
    assert ol.msm_g2(cf.G2_GEN, ol.le32(gamma)) == ol.g2_json(vk['vk_gamma_2'])
    assert ol.msm_g2(cf.G2_GEN, ol.le32(delta)) == ol.g2_json(vk['vk_delta_2'])


@pytest.mark.parametrize('nl', [10, 160])
def test_oracle_prover_equals_closed_form(nl):
    """oracle/groth16.c (NTT + Pippenger over the .zkey) against the field-only closed form: identical bytes for two voters and two (r, s) pairs"""
    (r1, zkp, vkp), seed = _artifacts(nl)
    zk = open(zkp, 'rb').read()
    for vs, (r, s) in ((1, (12345, 67890)), (2, (ol.R - 1, 1)))[:2 if nl == 10 else 1]:
        v = _voter(nl, vs, depth_c=min(nl, 13), depth_s=4)
        rc, w = ol.witness(v, nl); assert rc == 0
        rc, proof, pub = ol.prove(zk, w, r, s); assert rc == 0
        a, b, c = cf.proof_scalars(r1, seed, w, r, s)
        assert proof == cf.proof_from_scalars(ol, a, b, c)
    # and the closed form notices a wrong witness (the constraint check inside it)
    bad = bytearray(w); bad[32 * 40] ^= 1
    with pytest.raises(AssertionError):
        cf.proof_scalars(r1, seed, bytes(bad), 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize('nl', [10, 160])
def test_gpu_prover_equals_closed_form(nl):
    """the product's whole prove path (buildABC, transforms, five MSMs with constant folding, blinding) against the closed form, through the C ABI"""
    import torch  # noqa: F401
    import zkcensus_amd
    (r1, zkp, vkp), seed = _artifacts(nl)
    zk = open(zkp, 'rb').read()
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    voters = [_voter(nl, 5, depth_c=min(nl, 14), depth_s=min(nl, 9)), _voter(nl, 6, depth_c=min(nl, 160), depth_s=2)]      # the second: a leaf at the very bottom of the census tree (nothing folds there)
    ws, st = ctx.witness(voters, nLevels=nl)
    assert st == [0, 0]
    for w, (r, s) in zip(ws, ((3, 4), (ol.R - 2, 0x1234567890abcdef))):
        proof, pub = pk.prove(w, r, s)
        a, b, c = cf.proof_scalars(r1, seed, w, r, s)
        assert proof == cf.proof_from_scalars(ol, a, b, c), 'GPU proof differs from the toxic-waste closed form at nLevels = %d' % nl
    pk.close(); ctx.close()
