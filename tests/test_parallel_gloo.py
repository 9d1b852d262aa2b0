"""N > 1 path on CPU: two gloo ranks shard a census with parallel.shard_range, fabricate their proof records and gather them
with the same parallel.gather_records bench.py uses over RCCL."""
import os
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import zkcensus_amd
from zkcensus_amd import parallel


def _worker(rank, world, total, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = parallel.shard_range(rank, world, total)
    n = hi - lo
    proofs = b''.join(bytes([(v * 7 + k) % 251 for k in range(256)]) for v in range(lo, hi))
    pubs = b''.join(bytes([(v * 3 + k) % 241 for k in range(256)]) for v in range(lo, hi))
    rec = parallel.pack_records(proofs, pubs, [v % 7 for v in range(lo, hi)])
    allrec = parallel.gather_records(rec, world, dist, total)
    ok = allrec.shape == (total, parallel.record_width())
    for v in range(total):
        ok &= bytes(allrec[v, :256].tolist()) == bytes([(v * 7 + k) % 251 for k in range(256)])
        ok &= bytes(allrec[v, 256:512].tolist()) == bytes([(v * 3 + k) % 241 for k in range(256)])
        ok &= int(allrec[v, 512]) == v % 7
    t = torch.tensor([1.0 + rank]); dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the max-over-ranks timing reduction of bench.py
    ok &= float(t) == float(world)
    q.put((rank, bool(ok), n))
    dist.destroy_process_group()


def test_shard_ranges_cover():
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 7, 8, 1024, 8192, 8193):
            r = [parallel.shard_range(k, world, total) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_gather_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, 13, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs: p.join(60)
    assert res == [(0, True, 7), (1, True, 6)]
