import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, numpy as np
import zkcensus_amd
from zkcensus_amd import setup
ctx = zkcensus_amd.Context(0)
_, zp, _ = setup.ensure_test_artifacts(160)
pk = zkcensus_amd.ProvingKey(ctx, open(zp,'rb').read())
rng = np.random.default_rng(1)
n = pk.domain_size
sc = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint64).astype(np.uint32); sc[:, 7] &= 0x0fffffff
d = torch.from_numpy(sc.view(np.uint8).reshape(-1)).cuda()
for it in range(2):
    print('--- H msm, iteration', it, file=sys.stderr); sys.stderr.flush()
    pk.msm_debug(4, d.data_ptr(), n)
