"""synthetic large-census stress -- a G1 MSM over n = 2^20 bases k_i G with uniformly random 254-bit
scalars, and NTTs of 2^20 BN254 Fr elements, on one GPU.  (The zkCensus circuit itself cannot reach a 2^20 domain; nLevels = 252 is
covered by tests/test_gpu_prover.py::test_max_levels_nl252_config5.)

Checks, not just timings: the MSM result equals (sum s_i k_i mod r) G computed by the CPU oracle; inverse(forward(x)) == x byte for byte
and sampled forward outputs equal the oracle's transform.  Prints one JSON line.   usage: python tools/stress.py [logn=20] [reps=5]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import zkcensus_amd
from zkcensus_amd import engines
import oracle_lib as ol                      # checker only

R = ol.R


def rand_scalars(rng, n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) | (rng.integers(0, 2, size=(n, 4), dtype=np.uint64) << np.uint64(63))
    a[:, 3] &= np.uint64((1 << 61) - 1)      # < 2^253 < r: uniformly random 253-bit scalars
    return a


def as_ints(a):
    return [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 for x in a]


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n = 1 << logn
    ctx = zkcensus_amd.Context(0)
    rng = np.random.Generator(np.random.PCG64(1 << 20))
    # ---- MSM ----
    k = rand_scalars(rng, n); d_k = torch.from_numpy(k.view(np.uint8).reshape(-1).copy()).cuda()
    d_bases = torch.empty(64 * n, dtype=torch.uint8, device='cuda')
    t0 = time.perf_counter(); engines.g1_mul_batch(ctx, engines.G1_GENERATOR, d_k.data_ptr(), n, d_bases.data_ptr()); t_gen = time.perf_counter() - t0
    t0 = time.perf_counter(); tbl = engines.G1Bases(ctx, d_bases.data_ptr(), n); torch.cuda.synchronize(); t_load = time.perf_counter() - t0
    s = rand_scalars(rng, n); d_s = torch.from_numpy(s.view(np.uint8).reshape(-1).copy()).cuda()
    got = tbl.multiExpAffine(d_s.data_ptr())                     # warm-up + the checked result
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r2 = tbl.multiExpAffine(d_s.data_ptr()); ts.append(time.perf_counter() - t0)
        assert r2 == got
    t_chk = sum(a * b for a, b in zip(as_ints(s), as_ints(k))) % R
    msm_ok = got == ol.g1_mul(engines.G1_GENERATOR, t_chk)
    t_msm = min(ts)
    # ---- NTT ----
    nvec = 8
    v = rand_scalars(rng, n * nvec); d_v = torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).cuda()      # any residues < r, read as Montgomery form
    d_f = torch.empty_like(d_v); d_b = torch.empty_like(d_v)
    engines.fft(ctx, d_v.data_ptr(), d_f.data_ptr(), logn, nvec); torch.cuda.synchronize()
    tn = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); engines.fft(ctx, d_v.data_ptr(), d_f.data_ptr(), logn, nvec); torch.cuda.synchronize(); tn.append(time.perf_counter() - t0)
    engines.ifft(ctx, d_f.data_ptr(), d_b.data_ptr(), logn, nvec); torch.cuda.synchronize()
    ntt_roundtrip = bool(torch.equal(d_b, d_v))
    rinv = pow(1 << 256, -1, R)
    x0 = [w * rinv % R for w in as_ints(v[:n])]                   # vector 0 in standard form
    exp = ol.ntt(x0); gotf = d_f[:32 * n].cpu().numpy().tobytes()
    ntt_ok = all(int.from_bytes(gotf[32 * i:32 * i + 32], 'little') * rinv % R == exp[i] for i in list(range(0, n, max(1, n // 2048))) + [1, n - 1])
    t_ntt = min(tn) / nvec
    print(json.dumps({
        'config': 'SURVEY.md 8d config 5 (ii): synthetic stress, n = 2^%d, one MI355X' % logn,
        'msm_g1': {'n': n, 'window_bits': 17, 'ms': round(1e3 * t_msm, 3), 'points_per_s': round(n / t_msm), 'alg_GBps': round(n * 96 / t_msm / 1e9, 2),
                   'hbm_frac': round(n * 96 / t_msm / 8e12, 5), 'matches_oracle_exponent_space': msm_ok,
                   'bases_gen_s': round(t_gen, 2), 'table_build_s': round(t_load, 2), 'table_bytes': 15 * n * 64},
        'ntt': {'n': n, 'batch': nvec, 'ms_per_transform': round(1e3 * t_ntt, 3), 'alg_GBps': round(2 * n * 32 / t_ntt / 1e9, 1),
                'hbm_frac': round(2 * n * 32 / t_ntt / 8e12, 4), 'passes': 3, 'roundtrip_exact': ntt_roundtrip, 'matches_oracle_sampled': ntt_ok},
    }))
    tbl.close(); ctx.close()
    return 0 if (msm_ok and ntt_roundtrip and ntt_ok) else 1


if __name__ == '__main__':
    sys.exit(main())
