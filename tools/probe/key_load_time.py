import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch, zkcensus_amd
from zkcensus_amd import setup
_, zp, vp = setup.ensure_test_artifacts(160)
zk = open(zp,'rb').read()
ctx = zkcensus_amd.Context(0)
for i in range(3):
    t=time.perf_counter(); pk = zkcensus_amd.ProvingKey(ctx, zk); torch.cuda.synchronize(); print('load %.3f s' % (time.perf_counter()-t)); 
    free,total = torch.cuda.mem_get_info(); print('used GB %.2f' % ((total-free)/2**30)); pk.close()
