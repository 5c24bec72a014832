"""Multi-GPU sharding of a batch of independent trajectories.

Trajectories are independent optimisation problems, so the path shards embarrassingly:
rank r owns the contiguous block [r*B/G, (r+1)*B/G) of the global batch, solves it on its own
GPU with its own C-ABI handle, and the only exchange is ONE all-gather of the 16-byte
{final_cost f64, iterations i32, status i32} record per trajectory (RCCL over xGMI when the
backend is "nccl"; the same code runs on gloo for the CPU tests).
"""
import numpy as np

RECORD_BYTES = 16
RECORD_DTYPE = np.dtype([("final_objective", "<f8"), ("iterations", "<i4"), ("status", "<i4")])


def partition(global_batch, world, rank):
    """Contiguous block partition; the first (global_batch % world) ranks get one extra trajectory."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_records(results):
    """numpy structured results (pyapi.RESULT_DTYPE) -> uint8 array of 16-byte gather records."""
    rec = np.zeros(len(results), dtype=RECORD_DTYPE)
    rec["final_objective"] = results["final_objective"]
    rec["iterations"] = results["iterations"]
    rec["status"] = results["status"]
    return rec.view(np.uint8).reshape(-1)


def unpack_records(buf):
    return np.frombuffer(np.ascontiguousarray(buf).tobytes(), dtype=RECORD_DTYPE)


def shard_capacity(global_batch, world):
    """Records per rank in the gathered buffer: the largest shard of the block partition."""
    return -(-global_batch // world)


PAD_STATUS = -1   # status / iterations of a padding record (every byte 0xFF, as cddp_hip_allgather_results pads)


def pad_records(rec_local, capacity):
    """Pad a rank's uint8 record buffer (torch tensor) to `capacity` records; raises if the shard is larger."""
    import torch
    n = rec_local.numel() // RECORD_BYTES
    if rec_local.numel() % RECORD_BYTES or n > capacity:
        raise ValueError("shard of %d records (%d bytes) does not fit the gather capacity %d" % (n, rec_local.numel(), capacity))
    if n == capacity:
        return rec_local
    pad = torch.full(((capacity - n) * RECORD_BYTES,), 0xFF, dtype=rec_local.dtype, device=rec_local.device)
    return torch.cat([rec_local, pad])


def compact_records(gathered, global_batch, world):
    """Gathered world * capacity records (numpy uint8 / structured) -> the global_batch real records in batch order."""
    rec = unpack_records(gathered) if getattr(gathered, "dtype", None) != RECORD_DTYPE else gathered
    cap = shard_capacity(global_batch, world)
    if len(rec) != cap * world:
        raise ValueError("gathered buffer holds %d records, expected %d x %d" % (len(rec), world, cap))
    out = []
    for r in range(world):
        lo, hi = partition(global_batch, world, r)
        blk = rec[r * cap: r * cap + (hi - lo)]
        if np.any(blk["status"] == PAD_STATUS):
            raise ValueError("rank %d delivered padding inside its shard" % r)
        if np.any(rec[r * cap + (hi - lo):(r + 1) * cap]["status"] != PAD_STATUS):
            raise ValueError("rank %d: records beyond its shard are not padding" % r)
        out.append(blk)
    return np.concatenate(out)


def allgather_records(rec_local, world, dist=None, capacity=None):
    """rec_local: torch uint8 tensor of n_local*16 bytes (device tensor for nccl, cpu for gloo).
    The collective needs equally sized contributions: with `capacity` (= shard_capacity(global_batch, world)) every
    rank pads its shard to that many records (uneven block partitions); without it the shards must already be equal,
    which is ENFORCED (one all-reduce of the sizes) instead of assumed.
    Returns the gathered tensor of world*capacity*16 bytes -- the single collective of the path (torch.distributed
    flavour; bench.py uses the C-ABI's cddp_hip_allgather_results on RCCL instead)."""
    import torch
    if capacity is not None:
        rec_local = pad_records(rec_local, capacity)
    if world == 1 or dist is None:
        return rec_local
    if capacity is None:
        n = torch.tensor([rec_local.numel(), -rec_local.numel()], dtype=torch.int64, device=rec_local.device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        if int(n[0]) != -int(n[1]):
            raise ValueError("all_gather needs equal shards: sizes range from %d to %d bytes; pass capacity=" % (-int(n[1]), int(n[0])))
    out = torch.empty(rec_local.numel() * world, dtype=rec_local.dtype, device=rec_local.device)
    dist.all_gather_into_tensor(out, rec_local)
    return out
