"""N>1 path on CPU: world_size-2 gloo run of the sharding + single all-gather logic
(SURVEY.md section 8(e)).  Each rank solves its block of the batch with the CPU oracle (the GPU
handle is replaced only here, in the test, so that this runs without a GPU) and the gathered
16-byte records must equal a single-process solve of the whole batch."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_partition_blocks_cover_batch():
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    for B in (1, 7, 8, 4096, 16384, 32768):
        for G in (1, 2, 3, 4, 8):
            blocks = [sh.partition(B, G, r) for r in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            for (a, b), (c, d) in zip(blocks[:-1], blocks[1:]):
                assert b == c and b >= a
            assert max(b - a for a, b in blocks) - min(b - a for a, b in blocks) <= 1


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = _load("cddp_cpp_amd_pyapi", "cddp-cpp_amd/pyapi.py")
    _load("cddp_oracle_api", "oracle/oracle_api.py").attach(api)
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    p = api.pendulum_problem(api.SOLVER_IPDDP, True, horizon=40)
    B = 8
    x0 = api.batch_x0(p, B, 20260928, [0.1, 0.1])
    lo, hi = sh.partition(B, world, rank)
    res = api.oracle_solve_batch(p, x0[lo:hi], n_threads=1, want_traj=False)[0]
    rec = torch.from_numpy(sh.pack_records(res).copy())
    out = sh.allgather_records(rec, world, dist)
    dist.barrier()
    if rank == 0:
        q.put(out.numpy().tobytes())
    dist.destroy_process_group()


def test_two_rank_gloo_allgather_matches_single_process():
    import torch.multiprocessing as mp
    api = _load("cddp_cpp_amd_pyapi", "cddp-cpp_amd/pyapi.py")
    _load("cddp_oracle_api", "oracle/oracle_api.py").attach(api)
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    if not os.path.exists(api.ORACLE_LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    raw = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    got = sh.unpack_records(np.frombuffer(raw, dtype=np.uint8))
    p = api.pendulum_problem(api.SOLVER_IPDDP, True, horizon=40)
    x0 = api.batch_x0(p, 8, 20260928, [0.1, 0.1])
    ref = api.oracle_solve_batch(p, x0, n_threads=2, want_traj=False)[0]
    assert len(got) == 8
    assert np.array_equal(got["iterations"], ref["iterations"])
    assert np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["final_objective"], ref["final_objective"])


def test_uneven_shards_are_padded_and_enforced():
    """Strong scaling cuts a fixed global batch into blocks that may differ by one trajectory: the collective needs
    equal contributions, so shards are padded to shard_capacity (status -1 records) and compact_records drops the
    padding again; a shard that does not fit raises instead of silently corrupting the gather."""
    import torch
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    G, B = 3, 10
    cap = sh.shard_capacity(B, G)
    assert cap == 4
    parts = []
    for r in range(G):
        lo, hi = sh.partition(B, G, r)
        rec = np.zeros(hi - lo, dtype=sh.RECORD_DTYPE)
        rec["iterations"] = np.arange(lo, hi); rec["status"] = 1; rec["final_objective"] = 0.5 * np.arange(lo, hi)
        parts.append(sh.allgather_records(torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()), 1, None, capacity=cap))
    out = sh.compact_records(torch.cat(parts).numpy(), B, G)
    assert list(out["iterations"]) == list(range(B)) and np.all(out["status"] == 1)
    with pytest.raises(ValueError):
        sh.pad_records(torch.zeros(5 * sh.RECORD_BYTES, dtype=torch.uint8), cap)
    bad = torch.cat(parts).numpy().copy()
    bad[(cap - 1) * sh.RECORD_BYTES:(cap) * sh.RECORD_BYTES] = 0xFF      # rank 0's last REAL record overwritten by padding
    with pytest.raises(ValueError):
        sh.compact_records(bad, B, G)


def _worker_unequal(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    rec = torch.zeros((3 + rank) * sh.RECORD_BYTES, dtype=torch.uint8)
    try:
        sh.allgather_records(rec, world, dist)          # unequal shards without a capacity: must raise on EVERY rank
        ok = False
    except ValueError:
        ok = True
    out = sh.allgather_records(rec, world, dist, capacity=4)
    dist.barrier()
    q.put((rank, ok, int(out.numel())))
    dist.destroy_process_group()


def test_two_rank_gloo_unequal_shards():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unequal, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert got == [(0, True, 2 * 4 * 16), (1, True, 2 * 4 * 16)]


# ---- the same N > 1 flow with the PRODUCT handle on each rank (GPU box: both ranks share cuda:0) ---------------------
def _worker_hip(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = _load("cddp_cpp_amd_pyapi", "cddp-cpp_amd/pyapi.py")
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    p = api.cartpole_problem(api.SOLVER_IPDDP, True, 40)
    B = 131                                          # uneven: 66 + 65
    x0 = api.batch_x0(p, B, 20260928, [0.1, 0.3, 0.1, 0.1])
    lo, hi = sh.partition(B, world, rank)
    cap = sh.shard_capacity(B, world)
    hs = api.HipBatchSolver(p, hi - lo, device=0)
    hs.set_initial(np.ascontiguousarray(x0[lo:hi]))
    hs.solve()
    dev = torch.empty(cap * sh.RECORD_BYTES, dtype=torch.uint8, device="cuda")
    hs.allgather_results(None, 1, cap, dev.data_ptr())        # C-ABI: records + padding of this rank (world 1 = device copy)
    out = sh.allgather_records(dev.cpu(), world, dist)        # gloo carries the equal-sized padded blocks
    dist.barrier()
    if rank == 0:
        q.put(out.numpy().tobytes())
    hs.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_gloo_with_hip_handles():
    import torch.multiprocessing as mp
    api = _load("cddp_cpp_amd_pyapi", "cddp-cpp_amd/pyapi.py")
    sh = _load("cddp_sharding", "cddp-cpp_amd/sharding.py")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_hip, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    raw = q.get(timeout=600)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    got = sh.compact_records(np.frombuffer(raw, dtype=np.uint8), 131, 2)
    p = api.cartpole_problem(api.SOLVER_IPDDP, True, 40)
    x0 = api.batch_x0(p, 131, 20260928, [0.1, 0.3, 0.1, 0.1])
    hs = api.HipBatchSolver(p, 131); hs.set_initial(x0); hs.solve(); ref = hs.results(); hs.close()
    assert np.array_equal(got["iterations"], ref["iterations"]) and np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["final_objective"], ref["final_objective"])
