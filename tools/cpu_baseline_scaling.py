#!/usr/bin/env python3
"""Why does bench.py's cpu_baseline not use the 256 cores the GPU box shows?  ("multithreaded across all host cores of the GPU box".)

Measures the C oracle (tests/oracle_lib: witness + Groth16 prove at nLevels 160, one proof per thread, ctypes releases the GIL) at 4 .. 64 threads and records, beside every
level, what the kernel says about the CPU time this container may use:

  affinity             len(os.sched_getaffinity(0))            -- the cores the process may be SCHEDULED on (256 on the GPU boxes)
  cgroup cpu.max       quota / period                          -- the CPU TIME the container gets per period (cgroup v2; cpu.cfs_quota_us / cpu.cfs_period_us on v1)
  cgroup cpu.stat      nr_throttled, throttled_usec deltas      -- how often and for how long the level was throttled by that quota
  process CPU seconds  os.times() user + system deltas / wall   -- the cores' worth of CPU the level actually received

A level that asks for more threads than the quota allows receives `quota` cores' worth of time, spread over more threads: proofs/s stays flat while every proof takes longer.
That -- not the oracle's memory traffic, as DESIGN.md guessed in round 3 -- is the knee; bench.py now sizes the leg by min(affinity, quota) (`cores_quota`).

    python tools/cpu_baseline_scaling.py [--levels 4,8,16,32,64] [--out profiles/r04_cpu_baseline_scaling.json]       (host only; run it on the GPU box)"""
import argparse, json, os, random, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))


def cgroup_cpu():
    """-> dict(quota_cores or None, source, stat dict)"""
    out = {'quota_cores': None, 'source': None, 'stat': {}}
    try:
        rel = [l.strip().split(':', 2) for l in open('/proc/self/cgroup')]
    except OSError:
        rel = []
    v2 = [r[2] for r in rel if r[0] == '0']
    cands = []
    if v2:
        p = v2[0].strip('/')
        while True:                                  # the limit may sit on any ancestor
            cands.append(os.path.join('/sys/fs/cgroup', p))
            if not p:
                break
            p = os.path.dirname(p)
    for c in cands:
        f = os.path.join(c, 'cpu.max')
        if os.path.exists(f):
            q, per = open(f).read().split()
            if q != 'max':
                cores = int(q) / int(per)
                if out['quota_cores'] is None or cores < out['quota_cores']:
                    out['quota_cores'] = cores; out['source'] = f + ' = ' + q + ' ' + per
            st = os.path.join(c, 'cpu.stat')
            if not out['stat'] and os.path.exists(st):
                out['stat'] = {k: int(v) for k, v in (l.split() for l in open(st))}; out['stat_file'] = st
    for ctrl in ('cpu,cpuacct', 'cpu'):              # cgroup v1
        base = os.path.join('/sys/fs/cgroup', ctrl)
        q, per = os.path.join(base, 'cpu.cfs_quota_us'), os.path.join(base, 'cpu.cfs_period_us')
        if out['quota_cores'] is None and os.path.exists(q) and int(open(q).read()) > 0:
            out['quota_cores'] = int(open(q).read()) / int(open(per).read()); out['source'] = q
    return out


def quota_cores():
    c = cgroup_cpu()['quota_cores']
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--levels', default='4,8,16,32,64')
    ap.add_argument('--nlevels', type=int, default=160)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import oracle_lib as ol                      # the checker / baseline: this tool measures IT, nothing of the product runs here
    from census_gen import random_voter
    from zkcensus_amd import setup
    from concurrent.futures import ThreadPoolExecutor
    _, zkey_path, _ = setup.ensure_test_artifacts(a.nlevels)
    zk = open(zkey_path, 'rb').read()
    rng = random.Random(1)
    voter = random_voter(rng, ol.poseidon, nLevels=a.nlevels, depth_c=14, depth_s=13)
    ol.lib()

    def one(k):
        t0 = time.perf_counter()
        rc, w = ol.witness(voter, a.nlevels); assert rc == 0
        rc, p, pub = ol.prove(zk, w, 1 + k, 2 + k); assert rc == 0
        return time.perf_counter() - t0
    levels = []
    res = {'affinity': len(os.sched_getaffinity(0)), 'cpu_count': os.cpu_count(), 'cgroup': cgroup_cpu(), 'levels': levels}
    print(json.dumps({k: v for k, v in res.items() if k != 'levels'}), flush=True)
    for T in [int(x) for x in a.levels.split(',')]:
        c0 = cgroup_cpu()['stat']; t0 = os.times(); w0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            per = list(ex.map(one, range(T)))
        wall = time.perf_counter() - w0; t1 = os.times(); c1 = cgroup_cpu()['stat']
        cpu_s = (t1.user - t0.user) + (t1.system - t0.system)
        row = {'threads': T, 'proofs': T, 'wall_s': round(wall, 2), 'proofs_per_s': round(T / wall, 3), 'mean_seconds_per_proof': round(sum(per) / T, 2),
               'cpu_seconds': round(cpu_s, 1), 'cores_received': round(cpu_s / wall, 1),
               'throttled_periods': c1.get('nr_throttled', 0) - c0.get('nr_throttled', 0), 'throttled_ms': round((c1.get('throttled_usec', 0) - c0.get('throttled_usec', 0)) / 1e3)}
        levels.append(row); print(json.dumps(row), flush=True)
    if a.out:
        json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
