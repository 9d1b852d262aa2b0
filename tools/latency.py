#!/usr/bin/env python3
"""Single-proof latency (BASELINE configs[1]): inputs_example.json verbatim through the C ABI, per stage, on one MI355X."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import zkcensus_amd
from zkcensus_amd import setup

def main():
    ex = json.load(open(os.path.join(ROOT, 'tests/golden/ref/inputs_example.json')))
    _, zp, _ = setup.ensure_test_artifacts(160)
    t0 = time.perf_counter(); ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, open(zp, 'rb').read()); t_load = time.perf_counter() - t0
    flat = zkcensus_amd.flatten_inputs(ex)
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    nW = ctx.n_wires(160)
    d_w = torch.empty(nW * 32, dtype=torch.uint8, device='cuda'); d_s = torch.zeros(1, dtype=torch.int32, device='cuda')
    res = {}
    best = {}
    rs = lambda it: (11 + it).to_bytes(32, 'little') + (17 + it).to_bytes(32, 'little')
    for it in range(12):                                   # minimum over 12 rounds (the first ones grow work space and caches)
        t0 = time.perf_counter()
        ws, st = ctx.witness([flat])                       # host buffers in and out (2.6 MB witness over PCIe)
        t1 = time.perf_counter()
        pk.prove(ws[0], 11 + it, 17 + it)                  # host witness in, 256-byte proof out
        t2 = time.perf_counter()
        ctx.witness_dev(d_in.data_ptr(), 1, d_w.data_ptr(), d_s.data_ptr()); pk.prove_dev(d_w.data_ptr(), 11 + it, 17 + it)   # device-resident, two calls
        t3 = time.perf_counter()
        pk.fullprove_batch_dev(d_in.data_ptr(), 1, d_w.data_ptr(), d_s.data_ptr(), rs(it))                                  # device-resident, ONE call (inputs -> proof)
        t4 = time.perf_counter()
        for k, v in (('witness_host_ms', t1 - t0), ('prove_host_ms', t2 - t1), ('fullprove_device_resident_ms', t3 - t2), ('fullprove_one_call_ms', t4 - t3)):
            best[k] = min(best.get(k, 1e9), v)
    res = {k: round(v * 1e3, 2) for k, v in best.items()}; res['status'] = st[0]
    # per-stage device time of ONE device-resident fullProve (HIP events around each kernel category on the stream it is launched on; the G2 pass,
    # the blinding and the G1 pass run on three streams, so the categories overlap and do not add up to the end-to-end figure)
    cats = {0: 'witness', 1: 'buildABC_matvec', 2: 'ntt_joinABC', 3: 'msm_bucketing', 4: 'msm_accumulate_g1', 5: 'msm_accumulate_g2', 6: 'msm_reduce'}
    ctx._lib.zkc_profile_enable(ctx._h, 0x7f); torch.cuda.synchronize()
    ctx.witness_dev(d_in.data_ptr(), 1, d_w.data_ptr(), d_s.data_ptr()); pk.prove_dev(d_w.data_ptr(), 23, 29); torch.cuda.synchronize()
    stages = {}
    for cat, name in cats.items():
        ms, n, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        ctx._lib.zkc_profile_read(ctx._h, cat, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
        stages[name] = round(ms.value, 3)
    ctx._lib.zkc_profile_enable(ctx._h, 0)
    res['stage_ms_overlapped_not_additive'] = stages
    # the rapidsnark entry point a Go caller reaches through prover.Prove (zk_census_test.go:89): whole file images in, JSON out, key identified per call
    from zkcensus_amd import groth16
    zk = open(zp, 'rb').read(); wt = groth16.wtns.calculate(ex)
    pb, ub, err = ctypes.create_string_buffer(4096), ctypes.create_string_buffer(4096), ctypes.create_string_buffer(256)
    best = None
    for it in range(5):
        ps, us = ctypes.c_ulong(4096), ctypes.c_ulong(4096)
        t0 = time.perf_counter(); rc = ctx._lib.groth16_prover(zk, len(zk), wt, len(wt), pb, ctypes.byref(ps), ub, ctypes.byref(us), err, 256); dt = time.perf_counter() - t0
        assert rc == 0, err.value
        if it > 0:                                           # the first call loads the key into the entry point's own context
            best = dt if best is None else min(best, dt)
    res['groth16_prover_file_images_ms'] = round(best * 1e3, 2)
    res['key_load_and_precompute_s'] = round(t_load, 2)
    print(json.dumps(res))

if __name__ == '__main__':
    main()
