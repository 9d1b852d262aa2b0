#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rate of the C ABI's pool entry point: zkc_pool_fullprove_batch on 1024 voters of the synthetic census, host inputs in,
host proofs / signals / status out, next to the device-resident figure bench.py reports.  usage: pool_bench.py [B] [steps] [devices, e.g. 0 or 0,1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (loads torch's HIP runtime first, as _native.load does)
import zkcensus_amd
from zkcensus_amd import setup, census, groth16


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    devices = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '0').split(',')]
    _, zp, vp = setup.ensure_test_artifacts(160)
    ctx = zkcensus_amd.Context(devices[0])
    voters = census.synthetic_census(ctx, 8192, 160)[:B]
    flat = [zkcensus_amd.flatten_inputs(v, 160) for v in voters]
    pool = zkcensus_amd.DevicePool(devices, open(zp, 'rb').read())
    pool.fullprove_batch(flat, None)                                     # warm-up: work space, fold tables
    t0 = time.perf_counter()
    for _ in range(steps):
        proofs, pubs, st = pool.fullprove_batch(flat, None)
    dt = time.perf_counter() - t0
    assert not any(st)
    ok = bool(groth16.verify_batch(ctx, json.load(open(vp)), pubs, proofs))      # the product's batch verifier over all B proofs of the last step
    print(json.dumps({'entry_point': 'zkc_pool_fullprove_batch (host buffers in and out)', 'devices': devices, 'batch': B, 'steps': steps,
                      'proofs_per_s': round(B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 2), 'batch_verifier_all_valid': ok}))
    pool.close(); ctx.close()


if __name__ == '__main__':
    main()
