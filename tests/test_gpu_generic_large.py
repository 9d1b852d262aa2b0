"""BASELINE configs[4] as written: the generic prover -- groth16.prove(zkey, wtns) for a circuit that is NOT the census circuit (ts_inputs/src/example.ts:358 via fullProve;
rapidsnark's groth16_prover, zk_census_test.go:89) -- beyond the 2^17 domain of ZkFranchiseProofCircuit(160), towards the 2^20 ceiling of the reference's powers of tau
(circuit/circuit-compiler.sh:57).

What these sizes exercise that the census key never does: the five-kernel transform pair (domains above 2^18), 17-bit windows for witness sections of 2^16 wires and more,
level-1 bins that no longer fit the register path of the bucketing, the pass size cut down by the key's entry count (zkc_zkey_load), 10^5 .. 10^6-row jagged-diagonal matrices.

Two kinds of instance (tests/big_circuit.py): a random R1CS with a tenth of boolean wires (rows solved for a coefficient: one witness), and a circuit-shaped one whose every
constraint defines a wire, which has a witness for any inputs -- the batch tests prove DIFFERENT witnesses side by side.
Checked against the toxic-waste closed form (tests/closed_form.py: no NTT, no MSM, no .zkey), the pinned verifier, and -- stage by stage -- the C oracle's h evaluations.
ZKC_TEST_FULL=1 adds the 2^20 cases (minutes of host-side key generation); tools/generic_bench.py measures them (profiles/r04_generic_2p20.json)."""
import os
import time
import pytest
import oracle_lib as ol
import closed_form as cf
import big_circuit as bc
from test_generic_circuit import setup_key

FULL = os.environ.get('ZKC_TEST_FULL') == '1'


def _dev(torch, b):
    import numpy as np
    return torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize('logn', [19] + ([20] if FULL else []))
def test_transform_pair_above_2p18_matches_oracle(tmp_path, logn):
    """few wires, many rows: the key is cheap to make and the domain is what is under test (h = the 3 x (iNTT, coset shift, NTT) + joinABC of stage a3/a4)"""
    import torch
    import zkcensus_amd
    n_cons = (1 << logn) - (1 << (logn - 3)); n_wires = 3000
    r1 = str(tmp_path / 'wide.r1cs')
    w = bc.big_instance(r1, n_cons, n_wires, 2, seed=logn)
    zk, vk = setup_key(r1, 31337 + logn)
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    assert pk.domain_size == 1 << logn
    dw = _dev(torch, w)
    assert pk.debug_stage(dw.data_ptr(), 1) == ol.h_evals(zk, w), 'h evaluations on the odd coset differ from the oracle (domain 2^%d)' % logn
    # and the whole proof: closed form + verifier
    proof, pub = pk.prove(w, 11, 13)
    a, b, c = cf.proof_scalars(r1, 31337 + logn, w, 11, 13)
    assert proof == cf.proof_from_scalars(ol, a, b, c)
    assert ol.verify(vk, pub, proof)
    pk.close(); ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize('n_cons,n_in,n_pub', [(200000, 64, 4)] + ([(1000000, 64, 8)] if FULL else []))
def test_generic_prover_large_domain(tmp_path, n_cons, n_in, n_pub):
    """Full Groth16 proofs at a domain of 2^18 (2^20 with ZKC_TEST_FULL=1) for a circuit-shaped instance -- every constraint defines a wire, so any inputs have a witness -- and
    THREE DIFFERENT witnesses of it: sections of >= 2^16 wires take the 17-bit window, the pass holds fewer proofs.  Each proof alone, then the three in one batch, then the
    batch in passes of two; two of them against the closed form, all three through the pinned verifier."""
    import torch
    import zkcensus_amd
    import random
    t0 = time.time()
    r1 = str(tmp_path / 'chain.r1cs')
    n_wires = bc.chain_instance(r1, n_cons, n_in, n_pub, seed=n_cons)
    rng = random.Random(n_cons + 1)
    ws = [bc.chain_witness(n_cons, n_in, n_cons, [rng.getrandbits(253) if k else rng.getrandbits(1) for k in range(n_in)]) for _ in range(3)]
    assert len({w[-32:] for w in ws}) == 3                                   # different inputs, different last wires
    t1 = time.time()
    zk, vk = setup_key(r1, 777 + n_cons)
    t2 = time.time()
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    t3 = time.time()
    logn = (n_cons + n_pub).bit_length()
    assert pk.n_vars == n_wires and pk.n_public == n_pub and pk.domain_size == 1 << logn
    rs_int = [(ol.R - 5, 98765432109876543210), (8, 10), (3, ol.R - 1)]
    singles = [pk.prove(w, r, s) for w, (r, s) in zip(ws, rs_int)]
    t4 = time.time()
    for k, (w, (proof, pub)) in enumerate(zip(ws, singles)):
        assert pub == w[32:32 * (1 + n_pub)]
        if k < 2:
            a, b, c = cf.proof_scalars(r1, 777 + n_cons, w, *rs_int[k])
            assert proof == cf.proof_from_scalars(ol, a, b, c), 'GPU proof differs from the closed form at domain 2^%d (witness %d)' % (logn, k)
        assert ol.verify(vk, pub, proof)
    t5 = time.time()
    B = 3
    d_w = _dev(torch, b''.join(ws))
    rs = b''.join(int(x).to_bytes(32, 'little') for r_s in rs_int for x in r_s)
    proofs, pubs = pk.prove_batch_dev(d_w.data_ptr(), B, rs)
    assert [proofs[256 * k:256 * k + 256] for k in range(B)] == [p for p, _ in singles] and pubs == b''.join(u for _, u in singles)
    # the same batch in passes of two proofs (ZKC_INFLIGHT is read at key load): the pass loop, result-slot alternation and the buildABC prefetch at this size
    os.environ['ZKC_INFLIGHT'] = '2'
    try:
        pk2 = zkcensus_amd.ProvingKey(ctx, zk)
    finally:
        del os.environ['ZKC_INFLIGHT']
    proofs2, pubs2 = pk2.prove_batch_dev(d_w.data_ptr(), B, rs)
    assert proofs2 == proofs and pubs2 == pubs
    pk2.close()
    print('\n[generic 2^%d] instance + 3 witnesses %.1f s, setup %.1f s, key load %.1f s, three proofs %.2f s, closed form x 2 + verifier %.1f s' % (logn, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    pk.close(); ctx.close()


@pytest.mark.gpu
def test_generic_prover_half_of_the_wires_boolean(tmp_path):
    """Real witnesses are full of bits: half of the 150 000 wires here are 0 or 1, so digit 1 of the lowest window collects ~37 000 entries per section in ONE bucket (hundreds of
    segments: the heavy-bucket merge at a size the census key never reaches) while the zero wires drop out of every window.  One proof against the closed form, one batch."""
    import torch
    import zkcensus_amd
    n_cons, n_wires, n_pub = 200000, 150000, 4
    r1 = str(tmp_path / 'bits.r1cs')
    w = bc.big_instance(r1, n_cons, n_wires, n_pub, seed=4242, bool_frac=0.5)
    zk, vk = setup_key(r1, 909)
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    proof, pub = pk.prove(w, 21, 34)
    a, b, c = cf.proof_scalars(r1, 909, w, 21, 34)
    assert proof == cf.proof_from_scalars(ol, a, b, c) and ol.verify(vk, pub, proof)
    d_w = _dev(torch, w * 5)
    rs = b''.join(int(x).to_bytes(32, 'little') for k in range(5) for x in ((21, 34) if k == 0 else (k, k + 1)))
    proofs, _ = pk.prove_batch_dev(d_w.data_ptr(), 5, rs)
    assert proofs[:256] == proof and len({proofs[256 * k:256 * k + 256] for k in range(5)}) == 5
    pk.close(); ctx.close()
