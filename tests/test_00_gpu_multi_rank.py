"""GPU, N > 1: the multi-GPU path on whatever box the suite lands on.  On a one-GPU box these tests skip; on a box with several GPUs they run
`bench.py --gpus N` (N = min(visible GPUs, 4): one process per GPU, RCCL all_gather of the 513-byte records, every rank's proofs batch-verified and
compared with its block of the gathered records) as a child process, and the proving service / device pool over every visible GPU.
This file sorts first on purpose: bench.py's ranks are child processes started BEFORE this pytest process has initialised the GPU
(torch.cuda.device_count() does not initialise it on this image)."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    import torch
    return torch.cuda.device_count()


def test_bench_over_rccl_on_every_gpu_up_to_four():
    n = min(_ngpu(), 4)
    if n < 2:
        pytest.skip('one visible GPU: the N > 1 path is covered over gloo (tests/test_parallel_gloo.py, tests/test_bench_launcher.py)')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--batch', '96', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-extras'],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert j['n_gpus'] == n and j['scaling'] == 'weak' and j['value'] > 0
    v = j['verified']
    assert v['proofs'] == 96 * n and v['batch_verifier_all_valid'] and v['gathered_records_equal_per_rank_records'] and v['oracle_verifier_all_valid']


def test_service_and_pool_over_every_visible_gpu():
    n = min(_ngpu(), 8)
    if n < 2:
        pytest.skip('one visible GPU')
    import random, threading
    sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import oracle_lib as ol
    from census_gen import random_voter
    import zkcensus_amd
    from zkcensus_amd import setup
    nl = 10
    _, zp, vp = setup.ensure_test_artifacts(nl)
    zk = open(zp, 'rb').read(); vk = json.load(open(vp))
    rng = random.Random(8)
    voters = [random_voter(rng, ol.poseidon, nLevels=nl, depth_c=rng.randrange(2, 9), depth_s=rng.randrange(2, 9)) for _ in range(64 * n)]
    rs = b''.join(rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(2 * len(voters)))
    # pool: contiguous blocks over distinct devices, bytes equal the oracle's
    pool = zkcensus_amd.DevicePool(list(range(n)), zk)
    proofs, pubs, st = pool.fullprove_batch(voters, rs=rs, nLevels=nl)
    assert st == [0] * len(voters)
    for i in range(0, len(voters), 37):
        rc, w = ol.witness(voters[i], nl)
        rc2, op, ou = ol.prove(zk, w, int.from_bytes(rs[64 * i:64 * i + 32], 'little'), int.from_bytes(rs[64 * i + 32:64 * i + 64], 'little'))
        assert rc == 0 and rc2 == 0 and proofs[256 * i:256 * i + 256] == op and pubs[256 * i:256 * i + 256] == ou
    pool.close()
    # service: a long queue spills onto the other devices; every proof verifies
    os.environ['ZKC_SERVICE_SPILL'] = '8'
    try:
        svc = zkcensus_amd.ProvingService(list(range(n)))
    finally:
        del os.environ['ZKC_SERVICE_SPILL']
    out = [None] * len(voters)

    def caller(t):
        for i in range(t, len(voters), 64):
            out[i] = svc.fullprove(zk, voters[i], nLevels=nl)
    th = [threading.Thread(target=caller, args=(t,)) for t in range(64)]
    for t in th: t.start()
    for t in th: t.join()
    assert all(o[2] == 0 for o in out)
    for i in range(0, len(voters), 11):
        assert ol.verify(vk, out[i][1], out[i][0])
    st = svc.stats()
    assert st['requests'] == len(voters) and st['failed'] == 0 and st['devices'] == n and st['devices_used'] >= 1
    svc.close()
