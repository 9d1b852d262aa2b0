"""Where does one bench step (1024 proofs) spend host-visible time?  witness / rs bytes / prove_batch / pack."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import zkcensus_amd
from zkcensus_amd import setup, census, parallel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nl = 160
_, zp, vp = setup.ensure_test_artifacts(nl)
ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, open(zp, 'rb').read())
voters = census.synthetic_census(ctx, B, nl)
flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
d_inputs = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda(0)
nW = ctx.n_wires(nl)
d_wtns = torch.empty(B * nW * 32, dtype=torch.uint8, device='cuda'); d_status = torch.zeros(B, dtype=torch.int32, device='cuda')
rs = np.random.default_rng(1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.witness_dev(d_inputs.data_ptr(), B, d_wtns.data_ptr(), d_status.data_ptr(), nl); torch.cuda.synchronize(); t1 = time.perf_counter()
    rsb = b''.join(rs.bytes(31) + b'\0' for _ in range(2 * B)); t2 = time.perf_counter()
    p, pub = pk.prove_batch_dev(d_wtns.data_ptr(), B, rsb); t3 = time.perf_counter()
    rec = parallel.pack_records(p, pub, d_status.cpu().tolist()); t4 = time.perf_counter()
    print('B=%d witness %.1f ms, rs %.1f ms, prove_batch %.1f ms (%.0f proofs/s), pack %.1f ms, total %.1f ms -> %.0f proofs/s' % (
        B, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), B / (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0), B / (t4 - t0)))
