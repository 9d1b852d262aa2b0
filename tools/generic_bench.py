"""Measures the GENERIC prover -- groth16.prove(zkey, wtns) for a circuit that is not the census circuit -- at domains 2^14 .. 2^20 (BASELINE configs[4]: "~2^20-constraint
R1CS"; the ceiling of the reference's powers of tau, circuit/circuit-compiler.sh:57).  Per size: a circuit-shaped random R1CS (tests/big_circuit.py chain_instance: every constraint defines a wire), the test-only setup, key load,
one proof checked against the toxic-waste closed form and the pinned verifier (the checkers: tests/closed_form.py, the C oracle's verifier), then

  proofs/s        B proofs over four different witnesses with different (r, s) through prove_batch_dev (device-resident witnesses), pipelined on three streams
  stages          one call of the same B with every kernel on ONE stream (ZKC_SERIAL_STREAMS=1): isolated per-stage ms, algorithmic bytes and GB/s against 8 TB/s,
                  and for the G1 accumulation the mixed additions per second against the VALU capacity of profiles/r03_valu_model.json

    python tools/generic_bench.py [--logn 14,15,16,18,20] [--out profiles/r04_generic_2p20.json]
"""
import argparse, ctypes, json, os, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
HBM_PEAK_GBPS = 8000.0
CATS = {0: 'witness', 1: 'buildABC', 2: 'ntt_joinABC', 3: 'msm_bucketing', 4: 'msm_accumulate_g1', 5: 'msm_accumulate_g2', 6: 'msm_reduce', 7: 'msm_g1_streamed'}


def read_prof(ctx):
    prof = {}
    for cat, name in CATS.items():
        ms, n, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        ctx._lib.zkc_profile_read(ctx._h, cat, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
        prof[name] = {'ms': ms.value, 'launches': n.value, 'alg_bytes': by.value}
    return prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--logn', default='14,15,16,18,20')
    ap.add_argument('--out', default=None)
    ap.add_argument('--no-check', action='store_true', help='skip closed form / verifier (they are what takes minutes at 2^20)')
    args = ap.parse_args()
    import numpy as np, torch
    import zkcensus_amd
    import oracle_lib as ol, closed_form as cf, big_circuit as bc          # checkers only
    from test_generic_circuit import setup_key
    try:
        valu_cap = json.load(open(os.path.join(ROOT, 'profiles', 'r03_valu_model.json')))['capacity_madd_per_s']
    except Exception:
        valu_cap = 16.1e9
    ctx = zkcensus_amd.Context(0)
    results = []
    for logn in [int(x) for x in args.logn.split(',')]:
        n = 1 << logn
        n_cons = n - n // 16; n_in = 64; n_pub = 8; n_wires = 1 + n_in + n_cons
        B = {20: 24, 19: 48, 18: 64}.get(logn, 96)              # at least two passes at every size (zkc_zkey_load sizes a pass by the key's entry count: 12 proofs at 2^20)
        tmp = tempfile.mkdtemp(prefix='zkc_generic_')
        r1 = os.path.join(tmp, 'c.r1cs')
        # a circuit-shaped instance (every constraint defines a wire: any inputs have a witness) and NW different witnesses of it, tiled over the batch
        NW = 4
        t0 = time.time(); assert bc.chain_instance(r1, n_cons, n_in, n_pub, seed=logn) == n_wires
        import random
        rng = random.Random(logn)
        wits = [bc.chain_witness(n_cons, n_in, logn, [rng.getrandbits(253) for _ in range(n_in)]) for _ in range(NW)]
        w = wits[0]; t_inst = time.time() - t0
        t0 = time.time(); zk, vk = setup_key(r1, 2024 + logn); t_setup = time.time() - t0
        t0 = time.time(); pk = zkcensus_amd.ProvingKey(ctx, zk); torch.cuda.synchronize(); t_load = time.time() - t0
        assert pk.domain_size == n and pk.n_vars == n_wires
        r, s = ol.R - 7, 1234567890123456789
        proof, pub = pk.prove(w, r, s)
        checked = None
        if not args.no_check:
            t0 = time.time()
            a, b, c = cf.proof_scalars(r1, 2024 + logn, w, r, s)
            checked = {'equals_closed_form': proof == cf.proof_from_scalars(ol, a, b, c), 'verifier_accepts': bool(ol.verify(vk, pub, proof)), 'seconds': round(time.time() - t0, 1)}
            assert checked['equals_closed_form'] and checked['verifier_accepts'], checked
        d_w = torch.from_numpy(np.frombuffer(b''.join(wits), dtype=np.uint8).copy()).cuda().repeat((B + NW - 1) // NW)[:B * len(w)].contiguous()
        rs = b''.join(int(x).to_bytes(32, 'little') for k in range(B) for x in ((r, s) if k == 0 else (3 + 2 * k, 5 + 3 * k)))
        lat = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); p1, _ = pk.prove_batch_dev(d_w.data_ptr(), 1, rs[:64]); lat.append((time.perf_counter() - t0) * 1e3)
        assert p1 == proof
        proofs, _ = pk.prove_batch_dev(d_w.data_ptr(), B, rs)                 # warm-up (work space grows here)
        assert proofs[:256] == proof
        reps = 3
        ctx._lib.zkc_profile_enable(ctx._h, 0x10); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            pk.prove_batch_dev(d_w.data_ptr(), B, rs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        madds = read_prof(ctx)['msm_g1_streamed']['launches'] / (reps * B); ctx._lib.zkc_profile_enable(ctx._h, 0)
        pk.close()
        # isolated stages
        os.environ['ZKC_SERIAL_STREAMS'] = '1'
        try:
            pk_s = zkcensus_amd.ProvingKey(ctx, zk)
        finally:
            del os.environ['ZKC_SERIAL_STREAMS']
        ps, _ = pk_s.prove_batch_dev(d_w.data_ptr(), B, rs)
        same = ps == proofs
        ctx._lib.zkc_profile_enable(ctx._h, 0x7f); torch.cuda.synchronize(); t0 = time.perf_counter()
        pk_s.prove_batch_dev(d_w.data_ptr(), B, rs)
        torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
        prof = read_prof(ctx); ctx._lib.zkc_profile_enable(ctx._h, 0)
        pk_s.close()
        stages = {}
        for k, v in prof.items():
            if k == 'msm_g1_streamed' or v['ms'] <= 0:
                continue
            gbps = v['alg_bytes'] / (v['ms'] * 1e-3) / 1e9
            stages[k] = {'ms': round(v['ms'], 3), 'ms_per_proof': round(v['ms'] / B, 3), 'alg_bytes': v['alg_bytes'], 'GBps': round(gbps, 1), 'frac_of_8TBps': round(gbps / HBM_PEAK_GBPS, 4)}
        acc = stages.get('msm_accumulate_g1')
        if acc:
            acc['madds_per_s'] = round(madds * B / (acc['ms'] * 1e-3)); acc['valu_frac'] = round(acc['madds_per_s'] / valu_cap, 3)
        row = {'logn': logn, 'constraints': n_cons, 'wires': n_wires, 'public': n_pub, 'zkey_MB': round(len(zk) / 1e6, 1),
               'host_seconds': {'instance': round(t_inst, 1), 'setup': round(t_setup, 1), 'key_load': round(t_load, 2)},
               'checked': checked, 'single_proof_ms': round(min(lat), 2), 'batch': B, 'proofs_per_s': round(B / dt, 2), 'ms_per_proof': round(dt / B * 1e3, 2),
               'g1_madds_per_proof': round(madds), 'serial_call_ms': round(dt_s * 1e3, 1), 'serial_bytes_equal_pipelined': bool(same), 'stages_isolated': stages}
        print(json.dumps(row), flush=True)
        results.append(row)
        del d_w; torch.cuda.empty_cache()
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)
    out = {'what': 'generic Groth16 prover (groth16.prove(zkey, wtns), no constant folding, no witness generation) on random satisfiable R1CS instances; one MI355X; witnesses device-resident; '
                   'B proofs per call over four different witnesses of the instance, distinct (r, s)', 'sizes': results}
    if args.out:
        json.dump(out, open(args.out, 'w'), indent=1)
    ctx.close()


if __name__ == '__main__':
    main()
