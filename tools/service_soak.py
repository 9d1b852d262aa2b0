#!/usr/bin/env python3
"""Soak of the proving service under a load of MIXED batch sizes (the fault of DESIGN.md section 5 needed exactly this: calls of different pass shapes
in flight on one key).  T caller threads, each a loop of: think for a random time (0 .. 3 ms, sometimes 20 ms: the queue drains and refills, so batches of 1, 2, a few and
dozens of proofs alternate), then one zkc_service_fullprove or zkc_service_prove call on one of TWO keys of the circuit (two ceremonies), for `seconds` seconds.  Every proof
that came back is then checked by the batch verifier under the key it was asked for, in chunks; a chunk that fails is re-checked proof by proof so that the report names the
request.  A proof computed from another caller's inputs, another key, or a corrupted pass fails.

    python tools/service_soak.py [seconds=20] [threads=48] [nLevels=160]        -> one JSON object (profiles/r04_service_soak.json)"""
import json, os, random, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
SECONDS, T, NL = 20.0, 48, 160            # set from the command line below, or by an importer (tests/test_gpu_service.py)


def main():
    global SECONDS, T, NL
    import tempfile
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import setup, groth16
    from census_gen import random_voter
    import synth_voter
    poseidon = lambda xs: synth_voter.H(*xs)                     # pure-Python Poseidon of tools/: test data only
    le = lambda x: int(x).to_bytes(32, 'little')
    g1 = lambda p: le(p[0]) + le(p[1])
    g2 = lambda p: le(p[0][0]) + le(p[0][1]) + le(p[1][0]) + le(p[1][1])
    vk_bytes = lambda vk: g1(vk['vk_alpha_1']) + g2(vk['vk_beta_2']) + g2(vk['vk_gamma_2']) + g2(vk['vk_delta_2']) + b''.join(g1(q) for q in vk['IC'])
    _, z1, v1 = setup.ensure_test_artifacts(NL)
    with tempfile.TemporaryDirectory() as d:
        _, z2, v2 = setup.ensure_test_artifacts(NL, seed=77, directory=d)
        keys = [(open(z1, 'rb').read(), vk_bytes(json.load(open(v1)))), (open(z2, 'rb').read(), vk_bytes(json.load(open(v2))))]
    rng = random.Random(4)
    voters = [random_voter(rng, poseidon, nLevels=NL, depth_c=rng.randrange(3, 20), depth_s=rng.randrange(3, 20)) for _ in range(64)]
    flats = [zkcensus_amd.flatten_inputs(v, NL) for v in voters]
    ctx = zkcensus_amd.Context(0)
    ws, st = ctx.witness(voters, nLevels=NL)
    assert st == [0] * len(voters)
    svc = zkcensus_amd.ProvingService([0])
    for k in keys:
        svc.fullprove(k[0], flats[0], nLevels=NL)
    results = [[] for _ in range(T)]; errors = []
    stop = time.time() + SECONDS

    def caller(t):
        r = random.Random(1000 + t)
        try:
            while time.time() < stop:
                x = r.random()
                time.sleep(0.02 if x < 0.03 else r.random() * 0.003 if x < 0.7 else 0.0)
                k = r.randrange(2); i = r.randrange(len(voters))
                if r.random() < 0.8:
                    p, u, s = svc.fullprove(keys[k][0], flats[i], nLevels=NL); assert s == 0
                else:
                    p, u = svc.prove(keys[k][0], ws[i])
                results[t].append((k, i, p, u))
        except Exception as e:                       # noqa: BLE001 -- reported with the thread that saw it
            errors.append((t, repr(e)))
    th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
    t0 = time.time()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.time() - t0
    stats = svc.stats(); timing = svc.timing()
    svc.close()
    lib = ctx._lib
    bad = []; n = 0
    for k in (0, 1):
        recs = [(i, p, u) for r_ in results for (kk, i, p, u) in r_ if kk == k]
        n += len(recs)
        for c0 in range(0, len(recs), 512):
            chunk = recs[c0:c0 + 512]
            pr = b''.join(p for _, p, _ in chunk); pu = b''.join(u for _, _, u in chunk)
            if lib.zkc_verify_batch(ctx._h, keys[k][1], 8, pu, pr, len(chunk), None) != 1:
                for (i, p, u) in chunk:
                    if lib.zkc_verify_bin(keys[k][1], 8, u, p) != 1:
                        bad.append({'key': k, 'voter': i})
            # public signals belong to the voter that was asked for (nullifier = signal 2)
            for (i, p, u) in chunk:
                if int.from_bytes(u[64:96], 'little') != int(voters[i]['nullifier']):
                    bad.append({'key': k, 'voter': i, 'why': 'public signals of another voter'})
    ctx.close()
    print(json.dumps({'seconds': round(dt, 1), 'threads': T, 'nLevels': NL, 'proofs': n, 'proofs_per_s': round(n / dt, 1), 'batches': stats['batches'], 'largest_batch': stats['largest_batch'],
                      'mean_batch': round(n / max(1, stats['batches']), 2), 'key_loads': stats['key_loads'], 'evictions': timing['key_evictions'], 'caller_errors': errors[:5],
                      'proofs_that_failed_verification': bad[:10], 'all_valid': not bad and not errors}))
    return 0 if not bad and not errors else 1


if __name__ == '__main__':
    if len(sys.argv) > 1: SECONDS = float(sys.argv[1])
    if len(sys.argv) > 2: T = int(sys.argv[2])
    if len(sys.argv) > 3: NL = int(sys.argv[3])
    sys.exit(main())
