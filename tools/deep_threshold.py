#!/usr/bin/env python3
"""Where does a pass start to gain from the key's second section tables (zkc_zkey::c_deep)?  Voters whose leaves sit d levels down both trees, for several d, proved 188 at a
time through the census key: proofs/s and G1 additions per proof.  Run once with ZKC_DEEP_WIRES=1 (every pass of more than two proofs deep) and once with
ZKC_DEEP_WIRES=1000000000 (none) on ONE box and compare (tools/gpu/call.sh ... env:ZKC_DEEP_WIRES=1 py:tools/deep_threshold.py env:ZKC_DEEP_WIRES=1000000000 py:tools/deep_threshold.py).
    python tools/deep_threshold.py [depths=13,20,30,40,60,100,160]   -> one JSON line"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zkcensus_amd
from zkcensus_amd import census, setup, groth16

depths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '13,20,30,40,60,100,160').split(',')]
nl, B = 160, 188
_, zp, vp = setup.ensure_test_artifacts(nl)
zk = open(zp, 'rb').read(); vk = json.load(open(vp))
ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
nW = ctx.n_wires(nl); lib = ctx._lib
rs = os.urandom(1)  # placeholder, replaced below
import random
rng = random.Random(5)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rs = b''.join(rng.randrange(R).to_bytes(32, 'little') for _ in range(2 * B))
out = {'ZKC_DEEP_WIRES': os.environ.get('ZKC_DEEP_WIRES'), 'rows': []}
for d in depths:
    voters = census.deep_voters(ctx, B, nl, depth=d)
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    d_w = torch.empty(B * nW * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
    pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    assert int(d_st.abs().sum().item()) == 0
    lib.zkc_profile_enable(ctx._h, 0x10); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        p, u = pk.fullprove_batch_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), rs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, n, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    lib.zkc_profile_read(ctx._h, 7, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by)); lib.zkc_profile_enable(ctx._h, 0)
    out['rows'].append({'depth': d, 'proofs_per_s': round(3 * B / dt, 1), 'g1_additions_per_proof': round(n.value / (3 * B)), 'all_valid': bool(groth16.verify_batch(ctx, vk, u, p))})
print(json.dumps(out))
pk.close(); ctx.close()
