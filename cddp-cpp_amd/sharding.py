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


def allgather_records(rec_local, world, dist=None):
    """rec_local: torch uint8 tensor of n_local*16 bytes (device tensor for nccl, cpu for gloo).
    All ranks must hold equally sized shards (pad the global batch to a multiple of `world`).
    Returns the gathered tensor of world*n_local*16 bytes -- the single collective of the path."""
    import torch
    if world == 1 or dist is None:
        return rec_local
    out = torch.empty(rec_local.numel() * world, dtype=rec_local.dtype, device=rec_local.device)
    dist.all_gather_into_tensor(out, rec_local)
    return out
