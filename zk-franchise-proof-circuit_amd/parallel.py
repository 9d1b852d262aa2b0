"""Multi-GPU layout of the batch path: independent voter proofs, contiguous block split, one process per
GPU, no data-path collective; finished proofs (256 B proof + nPublic x 32 B signals + 1 status byte per voter) are
gathered with one all_gather (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch

PROOF_BYTES = 256


def shard_range(rank, world, total):
    """Voters [lo, hi) proved by `rank`: contiguous blocks, sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_width(n_public=8):
    return PROOF_BYTES + 32 * n_public + 1


def pack_records(proofs, publics, status, n_public=8):
    """bytes, bytes, list[int] -> uint8 tensor [B, 256 + 32*n_public + 1] (numpy views: this sits inside the timed step of bench.py)"""
    import numpy as np
    B = len(status)
    rec = np.empty((B, record_width(n_public)), dtype=np.uint8)
    rec[:, :PROOF_BYTES] = np.frombuffer(proofs, dtype=np.uint8).reshape(B, PROOF_BYTES)
    rec[:, PROOF_BYTES:-1] = np.frombuffer(publics, dtype=np.uint8).reshape(B, 32 * n_public)
    rec[:, -1] = np.asarray(status, dtype=np.int64).astype(np.uint8)
    return torch.from_numpy(rec)


def gather_records(local, world, dist=None, total=None, force_collective=False):
    """all_gather of the per-rank record tensors -> [total, width] in voter order.  Ranks may hold blocks that differ by one
    row (shard_range): blocks are padded to the largest one for the collective and trimmed afterwards.  force_collective runs the
    collective even for a single rank (the one-GPU test of the RCCL call path)."""
    if world == 1 and not force_collective:
        return local
    sizes = [shard_range(r, world, total)[1] - shard_range(r, world, total)[0] for r in range(world)] if total is not None else [local.shape[0]] * world
    m = max(sizes)
    pad = local if local.shape[0] == m else torch.cat([local, local.new_zeros(m - local.shape[0], local.shape[1])])
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    return torch.cat([o[:s] for o, s in zip(out, sizes)])
