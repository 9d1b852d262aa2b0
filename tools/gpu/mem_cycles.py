import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
import torch, numpy as np
import oracle_lib as ol
import zkcensus_amd
from zkcensus_amd import setup
from census_gen import random_voter
_, zp, _ = setup.ensure_test_artifacts(10)
zk = open(zp, 'rb').read()
rng = random.Random(3)
voters = [random_voter(rng, ol.poseidon, nLevels=10, depth_c=3, depth_s=2) for _ in range(70)]
flat = b''.join(zkcensus_amd.flatten_inputs(v, 10) for v in voters)
rs = b''.join(rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(140))
torch.cuda.init(); torch.cuda.synchronize()
def free(): torch.cuda.synchronize(); torch.cuda.empty_cache(); return torch.cuda.mem_get_info()[0]
for c in range(8):
    f0 = free()
    ctx = zkcensus_amd.Context(0); f1 = free()
    pk = zkcensus_amd.ProvingKey(ctx, zk); f2 = free()
    ws, st = ctx.witness(voters[:1], nLevels=10)
    pk.prove(ws[0], 3, 4); f3 = free()
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    d_w = torch.empty(70 * ctx.n_wires(10) * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(70, dtype=torch.int32, device='cuda')
    pk.fullprove_batch_dev(d_in.data_ptr(), 70, d_w.data_ptr(), d_st.data_ptr(), rs); f4 = free()
    pk.close(); f5 = free()
    ctx.close(); del d_in, d_w, d_st; f6 = free()
    print('cycle %d: start %.1f | ctx %+.1f | key %+.1f | prove %+.1f | batch %+.1f | key close %+.1f | ctx close %+.1f | net %+.1f MB' % (c, f0 / 1e6, (f1 - f0) / 1e6, (f2 - f1) / 1e6, (f3 - f2) / 1e6, (f4 - f3) / 1e6, (f5 - f4) / 1e6, (f6 - f5) / 1e6, (f6 - f0) / 1e6))
