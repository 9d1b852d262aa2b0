"""GPU: the stand-alone NTT and G1 MSM entry points (SURVEY.md 8d config 5 (ii) machinery) against the oracle.
NTT: bit-exact against the oracle's transform at 2^10, 2^17 (two passes) and 2^20 (three passes), plus the inverse round trip.
MSM: bases k_i G made on the GPU (sampled against the oracle's scalar multiplication), result against the oracle's MSM for a small n
and in exponent space ((sum s_i k_i) G) for n = 2^15 and 2^17 -- the check that also scales to the 2^20 stress run."""
import random
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
R = ol.R


def _dev(torch, b):
    import numpy as np
    return torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy()).cuda()


@pytest.fixture(scope='module')
def gpu():
    import torch, zkcensus_amd
    ctx = zkcensus_amd.Context(0)
    yield ctx, torch
    ctx.close()


@pytest.mark.parametrize('logn', [10, 17, 20])
def test_ntt_matches_oracle(gpu, logn):
    ctx, torch = gpu
    from zkcensus_amd import engines
    n = 1 << logn
    rng = random.Random(logn)
    vals = [rng.randrange(R) for _ in range(n)]
    vals[0], vals[1], vals[n - 1] = 0, R - 1, 1
    mont = b''.join((v * engines.R_MONT % R).to_bytes(32, 'little') for v in vals)
    src = _dev(torch, mont); dst = torch.empty_like(src); back = torch.empty_like(src)
    engines.fft(ctx, src.data_ptr(), dst.data_ptr(), logn)
    torch.cuda.synchronize()
    rinv = pow(engines.R_MONT, -1, R)
    got = dst.cpu().numpy().tobytes()
    exp = ol.ntt(vals)                                            # standard-form ints
    step = max(1, n // 4096)                                      # de-Montgomery in Python is the slow part: every element up to 2^12, a stride above
    for i in list(range(0, n, step)) + [1, n - 1]:
        assert int.from_bytes(got[32 * i:32 * i + 32], 'little') * rinv % R == exp[i], 'NTT element %d' % i
    engines.ifft(ctx, dst.data_ptr(), back.data_ptr(), logn)
    torch.cuda.synchronize()
    assert back.cpu().numpy().tobytes() == mont                   # inverse(forward(x)) == x, every byte


@pytest.mark.parametrize('logn', [8, 15, 17, 20])          # 2^20 = BASELINE configs[4] (ii), the synthetic large-census G1 MSM
def test_msm_over_generated_bases(gpu, logn):
    ctx, torch = gpu
    from zkcensus_amd import engines
    n = 1 << logn
    rng = random.Random(100 + logn)
    ks = [rng.randrange(1, R) for _ in range(n)]
    ks[0], ks[1] = 1, R - 1
    d_k = _dev(torch, b''.join(k.to_bytes(32, 'little') for k in ks))
    d_bases = torch.empty(64 * n, dtype=torch.uint8, device='cuda')
    engines.g1_mul_batch(ctx, engines.G1_GENERATOR, d_k.data_ptr(), n, d_bases.data_ptr())
    bases = d_bases.cpu().numpy().tobytes()
    for i in [0, 1, 2, n // 2, n - 1]:
        assert bases[64 * i:64 * i + 64] == ol.g1_mul(engines.G1_GENERATOR, ks[i]), 'k_i G, i = %d' % i
    tbl = engines.G1Bases(ctx, d_bases.data_ptr(), n)
    for trial in range(2):
        ss = [rng.randrange(R) for _ in range(n)]
        if trial == 1:                                            # witness-like: many zeros, ones and repeats
            for i in range(0, n, 3): ss[i] = rng.choice([0, 1, 1, 2, ss[0]])
        scb = b''.join(s.to_bytes(32, 'little') for s in ss)
        got = tbl.multiExpAffine(_dev(torch, scb).data_ptr())
        t = sum(s * k for s, k in zip(ss, ks)) % R
        assert got == ol.g1_mul(engines.G1_GENERATOR, t), 'exponent-space check, n = 2^%d' % logn
        if logn == 8:
            assert got == ol.msm_g1(bases, scb)
    tbl.close()


def test_engine_entry_points_reject_bad_input(gpu):
    """Error behaviour of the stand-alone entry points: a base off the curve or with a coordinate >= q is refused at load time
    (ZKC_ERR_FORMAT), and malformed NTT calls return ZKC_ERR_BAD_ARG instead of launching anything."""
    ctx, torch = gpu
    import zkcensus_amd
    from zkcensus_amd import engines
    n = 256
    good = b''.join(ol.g1_mul(engines.G1_GENERATOR, k + 1) for k in range(n))
    for bad_point in ((1).to_bytes(32, 'little') + (3).to_bytes(32, 'little'),            # (1, 3) is not on y^2 = x^3 + 3
                      b'\xff' * 32 + (2).to_bytes(32, 'little')):                          # x >= q
        buf = bytearray(good); buf[64 * 7:64 * 8] = bad_point
        with pytest.raises(zkcensus_amd.ZkcError) as ei:
            engines.G1Bases(ctx, _dev(torch, buf).data_ptr(), n)
        assert ei.value.code == 5                                                         # ZKC_ERR_FORMAT
    tbl = engines.G1Bases(ctx, _dev(torch, good).data_ptr(), n)                           # the clean set loads, infinity (all zero) included
    zero = bytearray(good); zero[0:64] = bytes(64)
    engines.G1Bases(ctx, _dev(torch, zero).data_ptr(), n).close()
    tbl.close()
    d = torch.zeros(32 * 1024, dtype=torch.uint8, device='cuda'); d2 = torch.zeros_like(d)
    for args in ((d.data_ptr(), d.data_ptr(), 10, 1), (d.data_ptr(), d2.data_ptr(), 2, 1), (d.data_ptr(), d2.data_ptr(), 10, 0)):
        with pytest.raises(zkcensus_amd.ZkcError) as ei:
            engines.fft(ctx, *args)
        assert ei.value.code == 4                                                         # ZKC_ERR_BAD_ARG
