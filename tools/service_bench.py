#!/usr/bin/env python3
"""The reference's call shape under load, measured: ONE voter per call, many callers at once.

  1. node napi/example.js ... voters.json   Promise.all over one groth16.fullProve per voter (ts_inputs/src/example.ts:358-362), 64 and 256 at once
  2. groth16_prover from T threads          the symbol go-rapidsnark binds (zk_census_test.go:89), whole .zkey / .wtns images per call
  3. zkc_service_fullprove from T threads   inputs in, proof out (what a cgo prover.Prove replacement calls)
All three end in the library's proving service (csrc/zkc_service.hip), which coalesces concurrent callers into pipeline passes.  Every proof of 2 and 3 is
checked by the batch verifier, every proof of 1 by groth16.verify inside the script.  usage: service_bench.py [threads] [calls per thread]  -> one JSON object.
node runs first, as a child started before this process touches the GPU."""
import ctypes, json, os, random, shutil, subprocess, sys, tempfile, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(NL)
    rng = random.Random(2024)
    voters = [random_voter(rng, poseidon, nLevels=NL, depth_c=rng.randrange(12, 18), depth_s=rng.randrange(12, 18)) for _ in range(T)]
    out = {'threads': T, 'calls_per_thread': PER, 'nLevels': NL}
    node = shutil.which('node')
    if node and os.path.exists(os.path.join(ROOT, 'napi', 'zkcensus.node')):
        with tempfile.TemporaryDirectory() as d:
            vp = os.path.join(d, 'voters.json'); json.dump(voters[:64], open(vp, 'w'))
            r = subprocess.run([node, os.path.join(ROOT, 'napi', 'example.js'), zkey_path, vkey_path, '-', vp], cwd=ROOT, capture_output=True, text=True, timeout=600)
            if r.returncode == 0:
                j = json.loads(r.stdout.strip().splitlines()[-1]); out['node_promise_all_fullProve'] = j['burst']; out['node_single_fullProve_ms_warm'] = j['msWarm']
            else:
                out['node_error'] = r.stderr[-400:]
    import torch  # noqa: F401
    import zkcensus_amd
    from zkcensus_amd import _native
    lib = _native.load()
    zk = open(zkey_path, 'rb').read(); vk = vk_bytes(json.load(open(vkey_path)))
    ctx = zkcensus_amd.Context(0)
    ws, st = ctx.witness(voters, nLevels=NL)
    assert st == [0] * T
    images = []
    for w in ws:
        n = lib.zkc_wtns_write(w, len(w) // 32, None, 0); buf = ctypes.create_string_buffer(n); lib.zkc_wtns_write(w, len(w) // 32, buf, n); images.append(buf.raw)
    flats = [zkcensus_amd.flatten_inputs(v, NL) for v in voters]
    svc = zkcensus_amd.ProvingService(default=True)

    def run(kind):
        res = [[] for _ in range(T)]

        def caller(t):
            for _ in range(PER):
                if kind == 'groth16_prover':
                    ps, us = ctypes.c_ulong(2048), ctypes.c_ulong(2048)
                    pb, ub, eb = ctypes.create_string_buffer(2048), ctypes.create_string_buffer(2048), ctypes.create_string_buffer(256)
                    rc = lib.groth16_prover(zk, len(zk), images[t], len(images[t]), pb, ctypes.byref(ps), ub, ctypes.byref(us), eb, 256)
                    assert rc == 0, eb.value
                    res[t].append((pb.value, ub.value))                      # JSON text; parsed after the clock stops (Python work under the GIL is not the prover's)
                else:
                    p, u, s = svc.fullprove(zk, flats[t], nLevels=NL); assert s == 0
                    res[t].append((p, u))
        caller(0); res[0].clear()                                  # key load, work space
        s0 = svc.stats(); tm0 = svc.timing(); th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        t0 = time.time()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.time() - t0; s1 = svc.stats(); tm1 = svc.timing()
        if kind == 'groth16_prover':
            res = [[(proof_bytes(json.loads(p)), b''.join(_le32(x) for x in json.loads(u))) for p, u in r] for r in res]
        proofs = b''.join(p for r in res for p, _ in r); pubs = b''.join(u for r in res for _, u in r); n = T * PER
        ok = lib.zkc_verify_batch(ctx._h, vk, 8, pubs, proofs, n, None)
        return {'proofs': n, 'seconds': round(dt, 4), 'proofs_per_s': round(n / dt, 1), 'batches': s1['batches'] - s0['batches'], 'largest_batch': s1['largest_batch'],
                'all_verified_by_batch_verifier': ok == 1, 'devices_used': s1['devices_used'],
                'worker_ms': {k: round((tm1[k] - tm0[k]) / 1e3, 1) for k in ('us_upload', 'us_wait_gpu', 'us_key', 'us_prove', 'us_finish')}}
    out['groth16_prover_threads'] = run('groth16_prover')
    out['service_fullprove_threads'] = run('fullprove')
    # [r4] two keys of the same circuit (two ceremonies: the reference keeps a key per environment, circuit/circuit-compiler.sh:15,82) hammered at once, half of the threads each:
    # with both resident (ZKC_SERVICE_KEYS >= 2, default 4) against one slot per device (every change of key frees 2 GB of tables and reloads for 0.6 s)
    def two_keys(slots):
        import tempfile
        os.environ['ZKC_SERVICE_KEYS'] = str(slots)
        try:
            svc2 = zkcensus_amd.ProvingService([0])
        finally:
            del os.environ['ZKC_SERVICE_KEYS']
        with tempfile.TemporaryDirectory() as d:
            _, z2, v2 = setup.ensure_test_artifacts(NL, seed=77, directory=d)
            zkB = open(z2, 'rb').read(); vkB = vk_bytes(json.load(open(v2)))
        keys = [(zk, vk), (zkB, vkB)]
        for k in keys:
            svc2.fullprove(k[0], flats[0], nLevels=NL)                   # both loaded once before the clock starts
        res = [[] for _ in range(T)]

        def caller(t):
            for _ in range(PER):
                p, u, s_ = svc2.fullprove(keys[t & 1][0], flats[t], nLevels=NL); assert s_ == 0
                res[t].append((p, u))
        if slots > 1:                                                    # an untimed round first: both keys' work space grows to a full pass (0.5 s each, once per key)
            th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
            for x in th: x.start()
            for x in th: x.join()
            for r_ in res: r_.clear()
        s0 = svc2.stats(); e0 = svc2.timing()['key_evictions']; th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        t0 = time.time()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.time() - t0; s1 = svc2.stats(); e1 = svc2.timing()['key_evictions']
        ok = True
        for par in (0, 1):
            pr = b''.join(p for t in range(par, T, 2) for p, _ in res[t]); pu = b''.join(u for t in range(par, T, 2) for _, u in res[t])
            ok = ok and lib.zkc_verify_batch(ctx._h, keys[par][1], 8, pu, pr, len(pr) // 256, None) == 1
        svc2.close()
        n = T * PER
        return {'resident_key_slots': slots, 'proofs': n, 'seconds': round(dt, 3), 'proofs_per_s': round(n / dt, 1), 'batches': s1['batches'] - s0['batches'],
                'key_loads_during_the_run': s1['key_loads'] - s0['key_loads'], 'evictions_during_the_run': e1 - e0, 'all_verified_under_their_own_key': bool(ok)}
    out['two_keys_half_the_threads_each'] = [two_keys(4), two_keys(1)]
    # one caller, one call at a time: the latency a lone sequential caller sees through the same entry points
    t0 = time.time(); [svc.fullprove(zk, flats[0], nLevels=NL) for _ in range(20)]; out['sequential_fullprove_ms'] = round((time.time() - t0) / 20 * 1e3, 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
