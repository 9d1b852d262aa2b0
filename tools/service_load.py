#!/usr/bin/env python3
"""The reference's call shape under load from NATIVE caller threads (tools/loadgen/loadgen.c: the way goroutines reach rapidsnark's groth16_prover through cgo,
zk_census_test.go:89), for one or several service configurations.

  usage: service_load.py [--threads 64,256] [--calls 16] [--configs "workers=4,pass=48;workers=2,pass=96"] [--modes prover,fullprove]
Every configuration makes its own service (the knobs are read at zkc_service_create) and loads its own key; every proof goes through the batch verifier.  One JSON line per
(configuration, mode, threads)."""
import argparse, ctypes, json, os, random, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
NL = 160


def loadgen():
    so = os.path.join(ROOT, 'tools', 'loadgen', 'libzkc_loadgen.so')
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.zkc_loadgen_run.argtypes = [ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                  ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return L


def _le32(x):
    return int(x).to_bytes(32, 'little')


def vk_bytes(vk):
    g1 = lambda p: _le32(p[0]) + _le32(p[1]) if int(p[2]) != 0 else bytes(64)
    g2 = lambda p: bytes(128) if int(p[2][0]) == 0 and int(p[2][1]) == 0 else _le32(p[0][0]) + _le32(p[0][1]) + _le32(p[1][0]) + _le32(p[1][1])
    return g1(vk['vk_alpha_1']) + g2(vk['vk_beta_2']) + g2(vk['vk_gamma_2']) + g2(vk['vk_delta_2']) + b''.join(g1(p) for p in vk['IC'])


def run_leg(lg, lib, svc, mode, threads, calls, zk, items, nvoters, vk, ctx):
    """mode: 'prover' (groth16_prover on the process-wide service), 'service_prove', 'fullprove'.  Returns the result dict."""
    n = threads * calls
    arr = (ctypes.c_char_p * nvoters)(*items); lens = (ctypes.c_size_t * nvoters)(*[len(x) for x in items])
    wall = ctypes.c_double(0); lat = (ctypes.c_double * n)()
    pj = uj = pr = pu = None; st = None
    if mode == 'prover':
        pj = ctypes.create_string_buffer(2048 * n); uj = ctypes.create_string_buffer(2048 * n)
        fn = ctypes.cast(lib.groth16_prover, ctypes.c_void_p); m = 0; h = None
    else:
        pr = ctypes.create_string_buffer(256 * n); pu = ctypes.create_string_buffer(256 * n); st = (ctypes.c_int32 * n)()
        fn = ctypes.cast(lib.zkc_service_fullprove if mode == 'fullprove' else lib.zkc_service_prove, ctypes.c_void_p); m = 1 if mode == 'fullprove' else 3; h = svc._h
    s0 = svc.stats(); t0 = svc.timing()
    failed = lg.zkc_loadgen_run(m, fn, h, threads, calls, zk, len(zk), None, 0, NL, 8, arr, lens, nvoters, pj, uj, pr, pu, st, ctypes.byref(wall), lat)
    s1 = svc.stats(); t1 = svc.timing()
    if mode == 'prover':
        proofs = b''; pubs = b''
        for i in range(n):
            p = json.loads(pj.raw[2048 * i:2048 * (i + 1)].split(b'\0', 1)[0]); u = json.loads(uj.raw[2048 * i:2048 * (i + 1)].split(b'\0', 1)[0])
            proofs += _le32(p['pi_a'][0]) + _le32(p['pi_a'][1]) + _le32(p['pi_b'][0][0]) + _le32(p['pi_b'][0][1]) + _le32(p['pi_b'][1][0]) + _le32(p['pi_b'][1][1]) + _le32(p['pi_c'][0]) + _le32(p['pi_c'][1])
            pubs += b''.join(_le32(x) for x in u)
    else:
        proofs = pr.raw; pubs = pu.raw
    ok = failed == 0 and lib.zkc_verify_batch(ctx._h, vk, 8, pubs, proofs, n, None) == 1
    ls = sorted(lat)
    nb = s1['batches'] - s0['batches']
    return {'mode': mode, 'threads': threads, 'calls_per_thread': calls, 'proofs': n, 'seconds': round(wall.value, 4), 'proofs_per_s': round(n / wall.value, 1), 'failed_calls': failed,
            'all_verified_by_batch_verifier': bool(ok), 'batches': nb, 'mean_batch': round(n / max(nb, 1), 1), 'largest_batch': s1['largest_batch'],
            'latency_ms': {'p50': round(ls[n // 2], 2), 'p95': round(ls[int(n * 0.95)], 2), 'max': round(ls[-1], 2)},
            'worker_ms_per_batch': {k: round((t1[k] - t0[k]) / 1e3 / max(nb, 1), 2) for k in ('us_upload', 'us_wait_gpu', 'us_key', 'us_prove', 'us_finish')}}


def make_voters(n, seed=2024):
    import synth_voter
    from census_gen import random_voter
    rng = random.Random(seed)
    H = lambda xs: synth_voter.H(*xs)                              # pure-Python Poseidon: test data only, nothing timed goes through it
    return [random_voter(rng, H, nLevels=NL, depth_c=rng.randrange(12, 18), depth_s=rng.randrange(12, 18)) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', default='64,256'); ap.add_argument('--calls', default='32,12'); ap.add_argument('--voters', type=int, default=64)
    ap.add_argument('--configs', default=''); ap.add_argument('--modes', default='service_prove,fullprove'); ap.add_argument('--rounds', type=int, default=1)
    a = ap.parse_args()
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import _native, setup
    lib = _native.load(); lg = loadgen()
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(NL)
    zk = open(zkey_path, 'rb').read(); vk = vk_bytes(json.load(open(vkey_path)))
    voters = make_voters(a.voters)
    ctx = zkcensus_amd.Context(0)
    ws, st = ctx.witness(voters, nLevels=NL)
    assert st == [0] * len(voters)
    images = []
    for w in ws:
        n = lib.zkc_wtns_write(w, len(w) // 32, None, 0); buf = ctypes.create_string_buffer(n); lib.zkc_wtns_write(w, len(w) // 32, buf, n); images.append(buf.raw)
    flats = [bytes(zkcensus_amd.flatten_inputs(v, NL)) for v in voters]
    configs = [c for c in a.configs.split(';') if c] or ['']
    for cfg in configs:
        env = {}
        for kv in cfg.split(','):
            if kv:
                k, v = kv.split('='); env[{'workers': 'ZKC_SERVICE_WORKERS', 'pass': 'ZKC_SERVICE_PASS', 'busy': 'ZKC_SERVICE_BUSY_WAIT_US', 'max': 'ZKC_SERVICE_MAX_BATCH'}.get(k, k)] = v
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            svc = zkcensus_amd.ProvingService(default=True) if cfg == 'default' else zkcensus_amd.ProvingService([0])
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        svc.fullprove(zk, flats[0], nLevels=NL)                    # key load, work space
        for mode in a.modes.split(','):
            items = flats if mode == 'fullprove' else ([w for w in ws] if mode == 'service_prove' else images)
            tl = [int(x) for x in a.threads.split(',')]; cl = [int(x) for x in a.calls.split(',')]
            for ti, T in enumerate(tl):
                calls = cl[min(ti, len(cl) - 1)]
                run_leg(lg, lib, svc, mode, T, 2, zk, items, len(voters), vk, ctx)          # untimed: staging buffers, pinned slots
                for _ in range(a.rounds):
                    r = run_leg(lg, lib, svc, mode, T, calls, zk, items, len(voters), vk, ctx)
                    r['config'] = cfg or 'defaults'; r['memory'] = svc.memory()
                    print(json.dumps(r), flush=True)
        if cfg != 'default':
            svc.close()


if __name__ == '__main__':
    main()
