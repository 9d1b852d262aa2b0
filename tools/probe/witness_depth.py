import os, sys, time, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import torch, numpy as np
import zkcensus_amd, oracle_lib as ol
from census_gen import random_voter
ctx = zkcensus_amd.Context(0)
nl = 160; nW = ctx.n_wires(nl)
rng = random.Random(1)
for d in (1, 4, 8, 16, 32, 64):
    v = random_voter(rng, ol.poseidon, nLevels=nl, depth_c=d, depth_s=d)
    flat = zkcensus_amd.flatten_inputs(v, nl)
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    d_w = torch.empty(nW * 32, dtype=torch.uint8, device='cuda'); d_s = torch.zeros(1, dtype=torch.int32, device='cuda')
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.witness_dev(d_in.data_ptr(), 1, d_w.data_ptr(), d_s.data_ptr(), nl); torch.cuda.synchronize()
        t1 = time.perf_counter()
    print('depth', d, 'witness_dev %.2f ms' % (1e3 * (t1 - t0)), 'status', d_s.cpu().tolist())
