"""bench.py --gpus N must launch its own ranks when it was not started by torch.distributed.run (the driver calls it both ways).
CPU rehearsal: --dry-run-cpu runs the same launcher, rendezvous, record gather and max-over-ranks reduction over gloo with
fabricated records; the real N > 1 run differs only in the backend (RCCL) and in where the records come from (the prover)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cmd):
    env = dict(os.environ); env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    j = _line([sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--dry-run-cpu'])
    assert j['n_gpus'] == 2 and j['dry_run'] and j['gathered_records_equal_per_rank_records'] and j['valid'] is False


def test_under_torch_distributed_run():
    j = _line([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(29700 + os.getpid() % 200), 'bench.py', '--gpus', '2', '--dry-run-cpu'])
    assert j['n_gpus'] == 2 and j['gathered_records_equal_per_rank_records']


def test_single_rank_dry_run():
    j = _line([sys.executable, 'bench.py', '--dry-run-cpu'])
    assert j['n_gpus'] == 1 and j['gathered_records_equal_per_rank_records']


def test_launcher_does_not_hang_when_a_rank_dies():
    """If one rank exits with an error the launcher terminates the others (they would wait in the collective until its time-out) and reports it."""
    import time
    env = dict(os.environ, ZKC_BENCH_TEST_FAIL_RANK='1'); env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--dry-run-cpu'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and 'rank 1 exited with status 7' in r.stderr and time.time() - t0 < 120


def test_eight_ranks_with_an_uneven_tail():
    """[r4] the driver's largest launch (BASELINE configs[3]: eight ranks) rehearsed on the CPU before an 8-GPU box ever sees it: eight gloo ranks, a census that does not
    divide by eight (ranks 3..7 prove one voter less), the census handed out by one broadcast from rank 0, one step time per rank in the line."""
    env_uneven = dict(os.environ, ZKC_BENCH_TEST_UNEVEN='1')
    env_uneven.pop('WORLD_SIZE', None); env_uneven.pop('RANK', None); env_uneven.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '2', '--warmup', '0', '--batch', '13', '--dry-run-cpu'], cwd=ROOT, env=env_uneven, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert j['n_gpus'] == 8 and j['voters'] == 8 * 13 - 5 and j['gathered_records_equal_per_rank_records']
    assert len(j['ms_per_step_per_rank']) == 8 and all(t >= 0 for t in j['ms_per_step_per_rank'])


def test_eight_ranks_one_dies():
    import time
    env = dict(os.environ, ZKC_BENCH_TEST_FAIL_RANK='5'); env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--dry-run-cpu'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and 'rank 5 exited with status 7' in r.stderr and time.time() - t0 < 120
