"""GPU: the collective bench.py uses for N > 1 (torch.distributed all_gather over the 'nccl' backend = RCCL on ROCm) executed with the only world
size a one-GPU box allows: one rank.  It proves that RCCL initialises on this stack and that parallel.gather_records drives it correctly with the
513-byte record tensors in HBM; the multi-rank behaviour of the same function is covered over gloo (tests/test_parallel_gloo.py) and by the
two-rank rehearsal on one GPU (profiles/r02_two_rank_rehearsal_one_gpu.json)."""
import os, socket
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_all_gather_of_records_one_rank():
    import torch, torch.distributed as dist
    import zkcensus_amd
    from zkcensus_amd import parallel
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    except Exception as e:                                      # no RCCL on this box: nothing of the product is at stake (the collective is torch's)
        pytest.skip('RCCL process group could not be created here: %s' % e)
    try:
        B = 1024
        proofs = bytes((7 * i) % 251 for i in range(256 * B)); pubs = bytes((3 * i) % 241 for i in range(256 * B))
        rec = parallel.pack_records(proofs, pubs, [i % 5 for i in range(B)])
        out = parallel.gather_records(rec.cuda(0), 1, dist, B, force_collective=True)
        assert out.is_cuda and out.shape == (B, parallel.record_width()) and bool((out.cpu() == rec).all())
        t = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the max-over-ranks reduction of bench.py
        assert float(t.item()) == 1.5
        dist.barrier()
    finally:
        dist.destroy_process_group()
