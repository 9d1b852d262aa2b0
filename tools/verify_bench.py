"""Batch verifier timing: N proofs of the nLevels=160 circuit, one zkc_verify_batch call against N zkc_verify_bin calls."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zkcensus_amd
from zkcensus_amd import setup, census, groth16, _native

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nl = 160
    _, zp, vp = setup.ensure_test_artifacts(nl)
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, open(zp, 'rb').read()); vk = json.load(open(vp))
    voters = census.synthetic_census(ctx, max(N, 64), nl)[:N]
    flat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in voters)
    d_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda()
    nW = ctx.n_wires(nl)
    d_w = torch.empty(N * nW * 32, dtype=torch.uint8, device='cuda'); d_s = torch.zeros(N, dtype=torch.int32, device='cuda')
    rs = np.random.default_rng(3).integers(0, 256, size=(2 * N, 32), dtype=np.uint8); rs[:, 31] = 0
    proofs, pubs = pk.fullprove_batch_dev(d_in.data_ptr(), N, d_w.data_ptr(), d_s.data_ptr(), rs.tobytes())
    assert int(d_s.abs().sum().item()) == 0
    vkb = groth16.vk_to_bytes(vk)
    t0 = time.perf_counter(); groth16.verify_batch(ctx, vkb, pubs, proofs, os.urandom(32)); first = time.perf_counter() - t0      # the key is made ready, the kernels load, the work space is allocated
    cpu0 = time.process_time(); t0 = time.perf_counter(); ok = groth16.verify_batch(ctx, vkb, pubs, proofs, os.urandom(32)); t1 = time.perf_counter(); cpu1 = time.process_time()
    lib = _native.load(); k = min(N, 64)
    t2 = time.perf_counter()
    for i in range(k):
        assert lib.zkc_verify_bin(vkb, 8, pubs[256 * i:256 * i + 256], proofs[256 * i:256 * i + 256]) == 1
    t3 = time.perf_counter()
    bad = bytearray(proofs); bad[256 * (N // 2) + 192:256 * (N // 2) + 256] = proofs[192:256]       # one proof gets another proof's C
    rej = groth16.verify_batch(ctx, vkb, pubs, bytes(bad), os.urandom(32))
    print(json.dumps({'N': N, 'batch_valid': ok, 'tampered_batch_rejected': not rej, 'batch_verify_s': round(t1 - t0, 4), 'first_call_s': round(first, 3), 'proofs_per_s_batch': round(N / (t1 - t0), 1),
                      'single_verify_ms': round(1e3 * (t3 - t2) / k, 2), 'speedup_vs_single': round((t3 - t2) / k * N / (t1 - t0), 1), 'host_threads': min(32, os.cpu_count()), 'host_cpu_s_batch': round(cpu1 - cpu0, 3)}))

if __name__ == '__main__':
    main()
