"""Helper of tests/test_00_gpu_switches.py (runs as a child process so that process-wide ZKC_* switches, which the library reads once, take effect): proves a fixed batch of
nLevels-10 voters -- inputs -> witness -> proof, 70 voters (two passes with ZKC_INFLIGHT=40: full-pass code paths from 32 proofs on) -- and one lone voter, with fixed (r, s), and
prints the SHA-256 of all proof and public-signal bytes."""
import hashlib, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
import zkcensus_amd
from zkcensus_amd import setup
from census_gen import random_voter
import synth_voter

nl, B = 10, 70
_, zp, vp = setup.ensure_test_artifacts(nl)
zk = open(zp, 'rb').read()
ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
rng = random.Random(20261004)
H = lambda xs: synth_voter.H(*xs)
voters = [random_voter(rng, H, nLevels=nl, depth_c=rng.randint(0, nl), depth_s=rng.randint(0, nl)) for _ in range(B)]
flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
rs = b''.join(rng.randrange(1 << 250).to_bytes(32, 'little') for _ in range(2 * B))
d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
d_w = torch.empty(B * ctx.n_wires(nl) * 32, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B, dtype=torch.int32, device='cuda')
h = hashlib.sha256()
ctx.witness_dev(d_in.data_ptr(), B, d_w.data_ptr(), d_st.data_ptr(), nLevels=nl)
assert int(d_st.abs().sum().item()) == 0
nofold_key = os.environ.get('ZKC_NO_FOLD') is not None           # such a key is not recognised as a census key: the witnesses are given (groth16.prove shape)
for b in (B, 1, 2, 40):
    p, u = pk.prove_batch_dev(d_w.data_ptr(), b, rs[:64 * b])
    h.update(p); h.update(u)
    if not nofold_key:                                          # inputs -> witness -> proof in one call: the same bytes
        d_w2 = torch.empty(b * ctx.n_wires(nl) * 32, dtype=torch.uint8, device='cuda')
        assert pk.fullprove_batch_dev(d_in.data_ptr(), b, d_w2.data_ptr(), d_st.data_ptr(), rs[:64 * b]) == (p, u)
print(json.dumps({'sha256': h.hexdigest()}))
pk.close(); ctx.close()
