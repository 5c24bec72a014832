"""Multi-GPU code paths that can run on the 1-GPU lease: the C-ABI's RCCL all-gather with a size-1 communicator, and
bench.py under torch.distributed.run with one rank (nccl backend, the communicator exchange, the device-side gather)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_allgather_size1_communicator(api):
    import torch
    p = api.pendulum_problem(api.SOLVER_IPDDP, True, 30)
    B = 70
    x0 = api.batch_x0(p, B, 20260928, [0.1, 0.1])
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0); hs.solve()
    ref = hs.results()
    comm = api.comm_init(api.comm_unique_id(), 1, 0, 0)          # ncclCommInitRank through the C-ABI
    try:
        for cap, c in ((B, comm), (B + 7, comm), (B + 7, None)):   # exact fit / padded, RCCL and the world-1 device copy
            dev = torch.zeros(cap * 16, dtype=torch.uint8, device="cuda")
            hs.allgather_results(c, 1, cap, dev.data_ptr())
            rec = np.frombuffer(dev.cpu().numpy().tobytes(), dtype=api.GATHER_DTYPE)
            assert np.array_equal(rec["iterations"][:B], ref["iterations"]) and np.array_equal(rec["status"][:B], ref["status"])
            assert np.array_equal(rec["final_objective"][:B], ref["final_objective"])
            assert np.all(rec["status"][B:] == -1) and np.all(rec["iterations"][B:] == -1)
        with pytest.raises(api.HipError):
            hs.allgather_results(comm, 1, B - 1, dev.data_ptr())    # capacity below the batch
        with pytest.raises(api.HipError):
            hs.allgather_results(None, 2, B, dev.data_ptr())        # NULL communicator with world > 1
    finally:
        api.comm_destroy(comm)
        hs.close()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_under_torchrun_one_rank(scaling):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: process group on nccl, unique-id broadcast,
    cddp_hip_comm_init, the RCCL all-gather inside the timed step -- everything of the N > 1 path except a second GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if scaling == "weak" else "29535", os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--workload", "pendulum", "--no-cpu-baseline", "--scaling", scaling] + (["--global-batch", "1000"] if scaling == "strong" else ["--batch", "512"])
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["scaling"] == scaling
    assert j["config"]["collective"].startswith("ncclAllGather")
    assert j["solve"]["gathered_records"] == j["config"]["global_batch"] == (1000 if scaling == "strong" else 512)
    assert j["value"] > 0 and j["roofline"]["frac"] > 0


def _run_bench(args, torchrun=False, port="29541"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):   # a clean, un-launched environment
        env.pop(k, None)
    bench = os.path.join(REPO, "bench.py")
    if torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", port, bench] + args
    else:
        cmd = [sys.executable, bench] + args
    return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)


def test_bench_gpus_beyond_the_node_fails_loudly():
    """`python bench.py --gpus N` with fewer than N devices must not print an N = 1 line under another
    name -- it exits non-zero with a message, with or without torchrun."""
    import torch
    n = torch.cuda.device_count() + 1
    out = _run_bench(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--workload", "pendulum", "--no-cpu-baseline"])
    assert out.returncode != 0
    assert "GPU(s) visible" in (out.stderr + out.stdout)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # under torchrun with ONE rank but --gpus 2 (a launcher / flag mismatch): refused as well
    if torch.cuda.device_count() >= 2:
        out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "pendulum", "--no-cpu-baseline"], torchrun=True)
        assert out.returncode != 0 and "does not match WORLD_SIZE" in (out.stderr + out.stdout)


def test_bench_plain_and_torchrun_single_gpu_agree():
    """`python bench.py --gpus 1` and the same under torch.distributed.run report the same job (same solve, same counters);
    only the collective differs (device copy vs size-1 ncclAllGather)."""
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "pendulum", "--batch", "512", "--no-cpu-baseline"]
    lines = []
    for tr in (False, True):
        out = _run_bench(args, torchrun=tr, port="29543")
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
    a, b = lines
    assert a["n_gpus"] == b["n_gpus"] == 1
    for k in ("mean_iterations", "status", "sweeps", "rollouts_useful", "gathered_records", "gathered_converged"):
        assert a["solve"][k] == b["solve"][k], k
    assert a["roofline"]["rollout_steps_credited"] == b["roofline"]["rollout_steps_credited"]
    assert a["config"]["collective"].startswith("none") and b["config"]["collective"].startswith("ncclAllGather")


def test_bench_default_line_carries_other_workloads():
    """the default single-GPU line also measures C2-CLDDP, C3 and the C4 / C5 per-GPU shares (3 steps each,
    outside the headline's timed region) so that the driver's record has them."""
    out = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ow = j["other_workloads"]
    assert len(ow) == 17 and not any("error" in w for w in ow), ow    # C2-CLDDP, resident LogDDP + MSIPDDP (f4, round 4), C3, C4 / C5 shares, 8 stack-fed sweeps + the plug-in solve (g1, round 6), 2 x MPC
    assert [w["solver"] for w in ow[:6]] == ["CLDDP", "LOGDDP", "MSIPDDP", "IPDDP", "IPDDP", "IPDDP"]
    for w in ow[:6]:
        assert w["value"] > 0 and 0 < w["roofline"]["frac"] < 1 and w["steps"] == 3
    for w in ow[6:14]:   # north_star's literal form: one launch over host-fed (N x batch) stacks, path rows and CLDDP, four shapes
        assert w["workload"].startswith("stack-fed sweep") and w["value"] > 0 and 0 < w["roofline"]["frac"] < 1
        assert all(c["sweeps_ok"] == c["batch"] for c in w["batch_curve"]) and w["roofline"]["algorithmic_bytes_per_launch"] == max(w["batch_curve"], key=lambda c: c["frac"])["stack_bytes"]
        assert w["roofline"]["traffic"] is None or 0.95 < w["roofline"]["traffic_over_algorithmic"] < 2.0   # counters of the same launch (profiles/r06_pmc_traffic_stackfed.json)
    # the form of the first (smallest) and of the last batch of each curve: small shapes with path rows start cooperative and end one-lane
    assert [w["batch_curve"][0]["form"] for w in ow[6:14]] == ["coop", "lane", "coop", "lane"] + ["coop"] * 4
    assert [w["batch_curve"][-1]["form"] for w in ow[6:14]] == ["lane"] * 4 + ["coop"] * 4
    assert ow[14]["workload"].startswith("host plug-in solve") and ow[14]["converged"] == ow[14]["batch"] and ow[14]["time_split"]["host_ms"] > 0
    for w in ow[15:]:   # the MPC re-solve lines (f1 caller side)
        assert w["workload"].startswith("MPC re-solves") and w["value"] > 0 and w["steps"] == 8 and len(w["iterations_by_round"]) == 8
        assert w["mean_iterations_per_resolve"] <= w["cold"]["mean_iterations"] + 5
    assert j["config"]["batch_per_gpu"] == 4096 and j["roofline"]["frac"] > 0


def test_bench_torchrun_line_carries_c4_c5_and_rccl_rank_count():
    """under torch.distributed.run the headline line is followed by BASELINE configs [3] / [4] (quadrotor,
    7-joint arm) measured through the same partition + all-gather path; with one rank it is the 8-GPU share on a size-1
    communicator -- the code path a world of 8 takes -- and the line reports the rank count RCCL itself sees."""
    out = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], torchrun=True, port="29547")
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["config"]["rccl_ranks"] == 1 and j["config"]["collective"].startswith("ncclAllGather")
    ow = j["other_workloads"]
    assert len(ow) == 2 and not any("error" in w for w in ow), ow
    assert "config[3]" in ow[0]["workload"] and "config[4]" in ow[1]["workload"]
    for w in ow:
        assert w["value"] > 0 and 0 < w["roofline"]["frac"] < 1 and w["n_gpus"] == 1 and w["rccl_ranks"] == 1


def test_comm_info_reports_what_rccl_sees(api):
    comm = api.comm_init(api.comm_unique_id(), 1, 0, 0)
    try:
        assert api.comm_info(comm) == (1, 0)
    finally:
        api.comm_destroy(comm)
