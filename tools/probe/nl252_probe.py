import os, sys, time, json, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import torch, numpy as np
import zkcensus_amd, oracle_lib as ol
from zkcensus_amd import setup
from census_gen import random_voter
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 252
t0 = time.time(); ctx = zkcensus_amd.Context(0)
_, zp, vp = setup.ensure_test_artifacts(nl); print('artifacts %.1f s' % (time.time() - t0), os.path.getsize(zp))
zk = open(zp, 'rb').read(); pk = zkcensus_amd.ProvingKey(ctx, zk); vk = json.load(open(vp))
print('nVars', pk.n_vars, 'domain', pk.domain_size)
rng = random.Random(5)
for depth in ((7, 5), (nl, nl)):
    v = random_voter(rng, ol.poseidon, nLevels=nl, depth_c=depth[0], depth_s=depth[1])
    rc, w = ol.witness(v, nLevels=nl); assert rc == 0
    gw = ctx.witness([v], nl)[0] if hasattr(ctx, 'witness') else None
    t1 = time.time(); p, pub = pk.prove(w, 111, 222); t2 = time.time()
    rc, op, opub = ol.prove(zk, w, 111, 222)
    print('depth', depth, 'gpu prove %.1f ms' % (1e3 * (t2 - t1)), 'parity', p == op and pub == opub, 'verify', ol.verify(vk, pub, p), 'witness parity', (gw == w) if gw is not None else 'n/a')
