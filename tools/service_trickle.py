#!/usr/bin/env python3
"""One submitting thread, asynchronous completions: the arrival pattern of `Promise.all(voters.map(v => groth16.fullProve(v, ...)))` in Node (one request
every `gap_us` microseconds from a single thread, completions delivered by callback), without Node.  Prints the rate and where the service's workers
spent their time (zkc_service_timing).  usage: service_trickle.py [requests] [gap_us] [rounds]"""
import ctypes, json, os, random, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
GAP = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NL = 160


def _le32(x):
    return int(x).to_bytes(32, 'little')


def _g1(p):
    return _le32(p[0]) + _le32(p[1]) if int(p[2]) != 0 else bytes(64)


def _g2(p):
    return bytes(128) if int(p[2][0]) == 0 and int(p[2][1]) == 0 else _le32(p[0][0]) + _le32(p[0][1]) + _le32(p[1][0]) + _le32(p[1][1])


def vk_bytes(vk):
    return _g1(vk['vk_alpha_1']) + _g2(vk['vk_beta_2']) + _g2(vk['vk_gamma_2']) + _g2(vk['vk_delta_2']) + b''.join(_g1(p) for p in vk['IC'])


def proof_bytes(pr):
    return _g1(pr['pi_a']) + _g2(pr['pi_b']) + _g1(pr['pi_c'])


def poseidon(xs):                                             # pure-Python Poseidon of tools/circuit_model.py: test data only, nothing timed goes through it
    import synth_voter
    return synth_voter.H(*xs)


def main():
    from census_gen import random_voter
    from zkcensus_amd import setup
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import _native
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(NL)
    rng = random.Random(7)
    base = [random_voter(rng, poseidon, nLevels=NL, depth_c=rng.randrange(12, 18), depth_s=rng.randrange(12, 18)) for _ in range(min(N, 64))]
    flats = [zkcensus_amd.flatten_inputs(v, NL) for v in base]
    lib = _native.load(); zk = open(zkey_path, 'rb').read(); vk = vk_bytes(json.load(open(vkey_path)))
    svc = zkcensus_amd.ProvingService(default=True)
    DONE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_char_p)
    out = []
    for rnd in range(ROUNDS):
        proofs = [ctypes.create_string_buffer(256) for _ in range(N)]; pubs = [ctypes.create_string_buffer(256) for _ in range(N)]
        left = [N]; ev = threading.Event(); rcs = []; lock = threading.Lock()

        def done(user, rc, status, text):
            with lock:
                rcs.append((rc, status)); left[0] -= 1
                if left[0] == 0:
                    ev.set()
        cb = DONE(done)
        s0, t0s = svc.stats(), svc.timing()
        t0 = time.perf_counter()
        for i in range(N):
            target = t0 + i * GAP * 1e-6
            while time.perf_counter() < target:
                pass
            rc = lib.zkc_service_submit_fullprove(svc._h, zk, len(zk), NL, flats[i % len(flats)], None, ctypes.cast(proofs[i], ctypes.c_char_p), ctypes.cast(pubs[i], ctypes.c_char_p),
                                                  ctypes.cast(cb, ctypes.c_void_p), None)
            assert rc == 0
        t_sub = time.perf_counter() - t0
        ev.wait(120)
        dt = time.perf_counter() - t0
        s1, t1s = svc.stats(), svc.timing()
        assert all(rc == 0 and st == 0 for rc, st in rcs) and len(rcs) == N
        ctx = zkcensus_amd.Context(0) if rnd == ROUNDS - 1 else None
        ok = lib.zkc_verify_batch(ctx._h, vk, 8, b''.join(p.raw for p in pubs), b''.join(p.raw for p in proofs), N, None) if ctx else None
        nb = s1['batches'] - s0['batches']
        out.append({'round': rnd, 'requests': N, 'gap_us': GAP, 'submit_ms': round(t_sub * 1e3, 1), 'total_ms': round(dt * 1e3, 1), 'proofs_per_s': round(N / dt, 1), 'batches': nb,
                    'worker_ms': {k: round((t1s[k] - t0s[k]) / 1e3, 1) for k in ('us_upload', 'us_wait_gpu', 'us_key', 'us_prove', 'us_finish')}, 'all_verified': ok})
        if ctx:
            ctx.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
