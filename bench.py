#!/usr/bin/env python3
"""bench.py -- batched CDDP solves on MI355X (BASELINE.json metric: trajectories/sec + HBM GB/s vs roofline).

A "step" is one complete batch solve (all DDP iterations, converged-or-max-iter) of the workload
BASELINE.json quotes the metric on: config[1], control-limited cart-pole (nx=4, nu=1, N=100),
batch 4096 random x0 per GPU, solved with the interior-point core (IPDDP, the north_star target) --
`--solver clddp` runs the BoxQP core instead.  Inputs (x0, U0) are resident in HBM before the timed
region.  One process per GPU; ranks shard independent trajectories (weak scaling) and exchange a
single RCCL all-gather of the 16-byte {cost, iterations, status} records per step.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import importlib.util
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))


def load_module(name, fn):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", fn))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_api():
    name = "cddp_cpp_amd_pyapi"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pyapi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle_api():
    """oracle/oracle_api.py: only the cpu_baseline leg below loads it."""
    name = "cddp_oracle_api"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "oracle", "oracle_api.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def algorithmic_bytes(nx, nu, N, m, ipddp):
    """bytes per trajectory of one derivative fill / backward sweep / rollout."""
    D = 8
    dyn = nx * nx + nx * nu
    cost = nx + nu + nx * nx + nu * nu + nu * nx
    con_in = 3 * m + m * nx + m * nu
    gain = nu * nx + nu
    val = nx + nx * nx
    con_out = 2 * m + 2 * m * nx
    b_fill = D * N * (nx + nu + dyn + cost + m + m * nx + m * nu)
    if ipddp:
        b_bwd = D * (N * (dyn + cost + con_in + gain + val + con_out) + val)
        b_fwd = D * N * (2 * (nx + nu) + gain + val + 2 * m + con_out + 3 * m + nx)
    else:
        b_bwd = D * (N * (dyn + cost + nx + nu + gain) + val)
        b_fwd = D * N * (2 * (nx + nu) + gain)
    return b_fill, b_bwd, b_fwd


def make_problem(api, workload, solver):
    sv = {"ipddp": api.SOLVER_IPDDP, "clddp": api.SOLVER_CLDDP, "logddp": api.SOLVER_LOGDDP, "msipddp": api.SOLVER_MSIPDDP}[solver]
    if workload == "cartpole_unc":
        p = api.cartpole_problem(sv, False)
        spread = [0.1, 0.3, 0.1, 0.1]
        desc = "cartpole nx=4 nu=1 N=100 UNCONSTRAINED (experiment), rk4, random x0"
    elif workload == "cartpole":
        p = api.cartpole_problem(sv, True)
        spread = [0.1, 0.3, 0.1, 0.1]
        desc = "cartpole nx=4 nu=1 N=100 control-limited (u in [-5,5]), rk4, random x0"
    elif workload == "unicycle":
        p = api.unicycle_problem(sv, 200, True)
        spread = [0.05, 0.05, 0.05]
        desc = "unicycle nx=3 nu=2 N=200 control box + ball obstacle (m=5), euler, random x0"
    elif workload == "quadrotor":   # config[3] per-GPU share: 16384 / 8 GPUs = 2048 trajectories
        p = api.quadrotor12_problem(sv, 400, True)
        spread = [0.02] * 12
        desc = "quadrotor (Euler-angle, nx=12 nu=4) N=400 thrust box, rk4, random x0"
    elif workload == "manip7":      # config[4] per-GPU share: 32768 / 8 GPUs = 4096 trajectories
        p = api.manipulator7_problem(sv, 150, True, 16)
        spread = [0.02] * 14
        desc = "7-joint manipulator nx=14 nu=7 N=150 torque box + terminal equality, 16-way parallel line search, rk4"
    elif workload == "pendulum":
        p = api.pendulum_problem(sv, True)
        spread = [0.1, 0.1]
        desc = "pendulum nx=2 nu=1 N=100 control-limited"
    else:
        raise SystemExit("unknown workload " + workload)
    return p, spread, desc


# committed PMC traffic summaries (profiles/make_traffic_json.py) per (workload, solver, batch)
TRAFFIC_FILES = {("cartpole", "ipddp", 4096): "r06_pmc_traffic.json", ("quadrotor", "ipddp", 2048): "r06_pmc_traffic_quadrotor.json",
                 ("manip7", "ipddp", 4096): "r06_pmc_traffic_manip7.json", ("cartpole", "clddp", 4096): "r06_pmc_traffic_clddp.json",
                 ("unicycle", "ipddp", 8192): "r06_pmc_traffic_unicycle.json", ("cartpole", "logddp", 4096): "r06_pmc_traffic_logddp.json",
                 ("pendulum", "msipddp", 4096): "r06_pmc_traffic_msipddp.json"}
ASSOC_ORDER_FILE = "r06_assoc_order.json"             # tests/test_cross_arithmetic.py::test_bench_batch_against_eigen_order_checker, collected by profiles/scripts/collect_parity_reports.py
CROSS_ARITHMETIC_FILE = "r05_cross_arithmetic.json"   # tests/test_cross_arithmetic.py's reports, copied from gpurun_out/
STRONG_GLOBAL_BATCH = {"cartpole": 4096, "cartpole_unc": 4096, "pendulum": 4096, "unicycle": 8192, "quadrotor": 16384, "manip7": 32768}
DEFAULT_BATCH = {"cartpole": 4096, "cartpole_unc": 4096, "pendulum": 4096, "unicycle": 8192, "quadrotor": 2048, "manip7": 4096}


def available_cpus():
    """CPUs this process may actually run on: the scheduler affinity mask capped by the cgroup CPU quota (v2 cpu.max, v1 cfs_quota).
    os.cpu_count() is the HOST's count, which a container lease does not own ('256 cores' scaling 6.7 x)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return n, {"os_cpu_count": os.cpu_count(), "affinity_cpus": aff, "cgroup_quota_cpus": quota}


def cpu_baseline(api, p, x0, U0, budget_s=20.0):
    """Oracle (CPU restatement, kind 'port') timed on the host cores of this box on a bounded sample (about budget_s seconds of wall time).

    Two builds of the same source are timed: the parity build's matrix capacity (every Mat / Vec carries 512 doubles) and a build with
    the capacity fitted to the workload's largest matrix (-DORACLE_MAT_CAP).  Each build walks a THREAD LADDER 1, 2, 4, ... up to the
    CPUs this process may use (available_cpus(): affinity capped by the cgroup quota, not os.cpu_count()), one work-queue run per
    rung on the first trajectories of the same batch (at least 8 per rung, so the single-thread figure is not a two-trajectory sample);
    `value` is the best rung of the faster build, `threads` the thread count of THAT rung and `cores` the CPUs the lease may use."""
    ncpu, cpu_info = available_cpus()
    oa = load_oracle_api()
    oa.attach(api)
    m = p.dual_dim()
    pT = sum(int(t.dim) for t in p._terms)
    side = max(p.nx, p.nu, m, pT + 1, 4)
    cap_fit = side * side + side          # the largest temporaries are side x side (+ one right-hand-side column)
    builds = {}
    for label, cap in (("capacity512", 512), ("fitted", cap_fit)):
        out = "/tmp/cddp_oracle_fast_%d_%s.so" % (os.getpid(), label)
        try:   # rebuilt for THIS box's CPU (-march=native)
            subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", "-DORACLE_MAT_CAP=%d" % cap, "-fPIC", "-shared", "-pthread", "-o", out,
                                   os.path.join(REPO, "oracle", "cddp_oracle.cpp")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            builds[label] = (out, cap)
        except Exception:
            pass
    ladder = []
    t = 1
    while t < ncpu:
        ladder.append(t); t *= 2
    ladder.append(ncpu)
    # the ladder may also probe beyond the quota-derived count once (oversubscription is visible as a flat rung, not hidden)
    if cpu_info["affinity_cpus"] > ncpu:
        ladder.append(min(cpu_info["affinity_cpus"], 2 * ncpu))

    def run(n, threads):
        n = int(min(x0.shape[0], max(1, n)))
        res, _, _, _, ms = api.oracle_solve_batch(p, x0[:n], None if U0 is None else U0[:n], n_threads=threads, fast=True, want_traj=False)
        return res, n, ms / 1e3

    results = {}
    res = None
    for label, (path, cap) in builds.items():
        oa.ORACLE_FAST_LIB_PATH = path
        oa._oracle_libs.pop(path, None)
        _, n_probe, s_probe = run(2, 1)                         # seconds per trajectory on one thread (2-trajectory probe, not reported)
        per_traj = max(s_probe / n_probe, 1e-4)
        per_rung = budget_s / len(builds) / (len(ladder) + 1)   # wall seconds per rung
        rungs = []
        for th in ladder:
            n = max(8, int(per_rung / per_traj) * th)           # >= 8 trajectories, about per_rung seconds if the rung scaled perfectly
            n = min(n, 64 * th)
            res, n_used, sec = run(n, th)
            rungs.append({"threads": th, "trajectories": n_used, "value": n_used / sec})
        single = rungs[0]["value"]
        best_rung = max(rungs, key=lambda r: r["value"])
        results[label] = {"value": best_rung["value"], "threads": best_rung["threads"], "sample_trajectories": best_rung["trajectories"],
                          "single_thread_value": single, "single_thread_trajectories": rungs[0]["trajectories"], "mat_capacity": cap,
                          "thread_scaling": best_rung["value"] / single, "thread_ladder": rungs}
    if not results:   # no compiler on the box: the committed parity build, one rung at the available CPU count
        n = min(x0.shape[0], 8 * ncpu)
        res, _, _, _, ms2 = api.oracle_solve_batch(p, x0[:n], None if U0 is None else U0[:n], n_threads=ncpu, fast=False, want_traj=False)
        results["parity_build"] = {"value": n / (ms2 / 1e3), "threads": ncpu, "sample_trajectories": n, "mat_capacity": 512}
    best = max(results, key=lambda k: results[k]["value"])
    rb = results[best]
    scaling = rb.get("thread_scaling")
    note = None
    if scaling is not None and rb["threads"] > 1 and scaling < 0.5 * rb["threads"]:
        note = ("best rung %d threads scales %.1f x over one thread (< half of the thread count): the rungs above are flat, i.e. the lease "
                "provides fewer physical cores than the affinity mask / quota advertises or the cores are shared SMT siblings; the checker itself "
                "is a lock-free work queue of independent solver objects (oracle/cddp_oracle.cpp::cddp_oracle_solve_batch)" % (rb["threads"], scaling))
    return {
        "value": rb["value"], "unit": "trajectories/s", "cores": ncpu, "threads": rb["threads"], "kind": "port",   # cores = CPUs the lease may use; threads = the best rung's thread count
        "sample": "first %d trajectories of the same batch on %d host threads (best rung of the ladder %s), oracle (Eigen-free CPU restatement, "
                  "-O3 -march=native, build '%s')" % (rb["sample_trajectories"], rb["threads"], [r for r in ladder], best),
        "single_thread_value": rb.get("single_thread_value"), "single_thread_trajectories": rb.get("single_thread_trajectories"),
        "thread_scaling": scaling, "cpus": dict(cpu_info, usable=ncpu), "scaling_note": note,
        "builds": results,
        "mean_iterations": float(np.mean(res["iterations"])),
    }


def parity_block():
    """What the parity claim rests on, with the cross-arithmetic flip rates of tests/test_cross_arithmetic.py (committed summary)."""
    out = {"status": "checked against the CPU restatement (oracle/ + its numpy twin), not against a cddp-cpp binary: parity unpinned (DESIGN.md 5)",
           "strict": "every -m gpu comparison runs the checker in the library's own sin / cos / log / pow (common-mode in those routines): status, "
                     "iterations, sweeps, rollouts identical for every trajectory of the benchmarked batches (tests/test_full_size.py)"}
    try:
        cj = json.load(open(os.path.join(REPO, "profiles", CROSS_ARITHMETIC_FILE)))
        out["cross_arithmetic_vs_glibc_checker"] = {
            w: {"compared": d["compared"], "count_flip_frac": d["count_flip_frac"], "work_flip_frac": d["work_flip_frac"],
                "objective_1e-7_mismatch_frac": d["objective_1e-7_mismatch_frac"], "yardstick": d.get("yardstick")} for w, d in cj["workloads"].items()}
        out["cross_arithmetic_source"] = "profiles/%s (%s)" % (CROSS_ARITHMETIC_FILE, cj["source"])
    except Exception:
        out["cross_arithmetic_vs_glibc_checker"] = None
    # round 6: summation ORDER -- the library (serial sums) against the checker in Eigen 3.4's SSE2 packet order as restated in
    # oracle/linalg.hpp::assoc_mode (a reading of Eigen's kernels, not a measurement: no Eigen in the build image)
    try:
        aj = json.load(open(os.path.join(REPO, "profiles", ASSOC_ORDER_FILE)))
        out["summation_order_vs_eigen_packet_model"] = {
            w: {"compared": d["compared"], "count_flip_frac": d["count_flip_frac"], "work_flip_frac": d["work_flip_frac"],
                "objective_1e-7_mismatch_frac": d["objective_1e-7_mismatch_frac"], "bitwise_equal_objective_frac": d["bitwise_equal_objective_frac"]} for w, d in aj["workloads"].items()}
        out["summation_order_source"] = "profiles/%s (%s)" % (ASSOC_ORDER_FILE, aj["source"])
    except Exception:
        out["summation_order_vs_eigen_packet_model"] = None
    return out


def roofline_block(api, p, m, B, workload, solver, st, prof, stats, sweep_dominates, concurrency=1, groups=1):
    """`roofline` object of one workload: SURVEY 8(d) algorithmic bytes of the dominant kernel class / its hipEvent time.

    st = stats of the last timed step, prof = stats of the untimed solve with every class bracketed, stats = all timed steps
    (they bracket the dominant class only)."""
    ipddp = solver in ("ipddp", "msipddp")   # MSIPDDP: the interior-point byte model (its costate / defect rows are not credited)
    b_fill, b_bwd, b_fwd = algorithmic_bytes(p.nx, p.nu, p.N, m, ipddp)
    # the dominant class: hipEvent time inside the timed steps; the others: from the profiling solve
    bwd_ms = float(np.mean([s.backward_ms for s in stats])) if sweep_dominates else float(prof.backward_ms)
    fwd_ms = float(prof.forward_ms) if sweep_dominates else float(np.mean([s.forward_ms for s in stats]))
    upd_ms = float(prof.update_ms)
    solve_ms = float(np.mean([s.solve_ms for s in stats]))
    bytes_bwd = b_fill * st.traj_iterations + b_bwd * st.sweeps
    # exact credit: the steps the walked trials actually traversed (a trial the reference abandons at its first
    # fraction-to-boundary violation, ipddp_solver.cpp:1632-1645, is credited the steps completed before it, not N)
    bytes_fwd = (b_fwd / p.N) * st.rollout_steps
    bytes_fwd_full_rollouts = b_fwd * st.rollouts   # round-1 accounting (every walked trial credited N steps), for comparison
    gbps_bwd = bytes_bwd / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
    gbps_fwd = bytes_fwd / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0
    gbps_all = (bytes_bwd + bytes_fwd) / (solve_ms * 1e-3) / 1e9
    lean = solver == "ipddp" and m > 0
    if sweep_dominates:
        if workload == "manip7" and ipddp:
            sweep_label = "k_derivs+k_te_condense+k_backward_te_coop+k_te_post"
        elif lean:
            # (nx <= 8: the derivative fill is fused into k_condense<.., true>, launch.hpp::derivs -- there is no k_derivs launch)
            # (round 6, nx <= 8: ONE launch -- the role-split sweep's helper wavefronts do k_condense's and k_post's work, kernels_coop.hpp)
            sweep_label = "k_derivs+k_condense+k_backward_ipddp_coop_big2+k_post" if p.nx > 8 else "k_backward_ipddp_coop"
        elif solver == "msipddp":  # resident MSIPDDP (kernels_msipddp.hpp): the split path-constrained sweep (round 5), the fused one-lane kernel otherwise
            # (path rows, nx <= 8: the derivative fill rides in k_ms_condense<.., true> -- no k_derivs launch)
            sweep_label = (("k_ms_condense+k_backward_msipddp_lean+k_ms_post" if p.nx <= 8 else "k_derivs+k_ms_condense+k_backward_msipddp_lean+k_ms_post")
                           if m > 0 else "k_derivs+k_backward_msipddp")
        elif solver == "logddp":   # resident LogDDP: the LogDDP mode of the cooperative sweep up to nx = 8, scored with CLDDP's byte model (the
            sweep_label = "k_derivs+k_backward_coop_plain" if p.nx <= 8 else "k_derivs+k_backward_logddp"   # barrier rows it also reads are not credited)
        else:
            sweep_label = "k_derivs+k_backward_coop_plain"
        dom = (sweep_label, gbps_bwd, bwd_ms, bytes_bwd)
        pmc_key = None
    else:
        dom = ("k_forward_ipddp_pc" if lean else ("k_forward_%s_pc" % solver if solver in ("msipddp", "logddp") else "k_forward_%s" % solver), gbps_fwd, fwd_ms, bytes_fwd)
        pmc_key = dom[0]
    PEAK = 8000.0   # GB/s HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy
    n_launch = max(1, st.outer_iterations)
    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure
    # is the per-launch FETCH_SIZE/WRITE_SIZE mean of the committed rocprofv3 --pmc passes over this very command
    # (profiles/r0N_pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes), null for other workloads.
    traffic = None
    traffic_note = None
    counter_whole = None
    tfile = TRAFFIC_FILES.get((workload, solver, B))
    if tfile:
        try:
            pj = json.load(open(os.path.join(REPO, "profiles", tfile)))
            # counter-based whole-solve bandwidth: every kernel's FETCH x 2 + WRITE bytes of one solve / the solve's time
            tot = float(sum(v["bytes_per_solve"] for v in pj["kernels"].values()))
            counter_whole = {"bytes_per_solve": tot, "GBps": tot / (solve_ms * 1e-3) / 1e9, "frac": tot / (solve_ms * 1e-3) / 1e9 / PEAK,
                             "source": "profiles/%s, all kernels" % tfile}
            # per launch of the dominant kernel class, like `algorithmic_bytes_per_launch`: the counter bytes of every kernel of the
            # class over one solve / that solve's outer iterations (an iteration is one sweep, and one or two rollout launches
            # depending on the ladder shape the solver picked)
            traffic = sum(pj["kernels"][k]["bytes_per_solve"] for k in dom[0].split("+")) / n_launch
            traffic_note = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, summed over the launches of %s in one solve "
                            "-- %d solves in the profiled command, counted from its k_init dispatches -- / outer iterations)" % (tfile, dom[0], pj["solves_in_profile"]))
        except Exception:
            traffic = None
    # classes whose SURVEY 8(d) MODEL rate exceeds the HBM peak: the model charges bytes the kernel serves from cache (e.g. K / k per alpha
    # of a best-merit ladder) -- such a figure is a model artefact, not bandwidth; the counter-based figure beside it is the physical one
    exceeds = [name for name, g in (("backward", gbps_bwd), ("forward", gbps_fwd), ("whole_solve", gbps_all)) if g > PEAK]
    return {
        "bound": "hbm", "kernel": dom[0], "achieved": dom[1], "peak": PEAK, "unit": "GB/s", "frac": dom[1] / PEAK,
        # the two readings of `frac` under the static CU partition: `frac_paired` = the concurrent launches of all tile
        # groups together against the whole chip's peak (= frac); `frac_per_kernel_on_its_partition` = ONE group's launch (what a rocprof
        # per-kernel average times) against the whole chip's peak, i.e. frac / groups in flight.  The overlap itself is kept as a kernel-trace
        # excerpt with both groups' start / end stamps: profiles/r06_group_overlap.md
        "frac_paired": dom[1] / PEAK, "frac_per_kernel_on_its_partition": dom[1] / PEAK / max(1, concurrency),
        "concurrent_groups": concurrency, "tile_groups": groups,
        "launch_definition": ("one launch = the %d concurrent half-batch launches of the static CU partition (each group on its own half of the CUs, "
                              "cddp_hip_concurrency); class times are the mean over the concurrent groups' streams, so a per-kernel rocprof "
                              "average is the duration of ONE of the %d concurrent launches" % (concurrency, concurrency)) if concurrency > 1 else "one launch = one kernel launch over the whole batch",
        "model_rate_exceeds_peak": exceeds, "whole_solve_counter_based": counter_whole,
        "frac_of_measured_copy_6290": dom[1] / 6290.0, "traffic": traffic, "traffic_source": traffic_note,
        "algorithmic_bytes_per_launch": dom[3] / n_launch, "avg_launch_ms": dom[2] / n_launch,
        "launches": n_launch,
        "algorithmic_bytes_note": "SURVEY 8(d) bytes per rollout STEP x the steps the reference's line search traverses: the trials the "
                                  "selection rule walks (first-success: up to the winner; best-merit: the whole ladder), each credited the steps "
                                  "completed before it is abandoned (fraction-to-boundary violation), N if it runs through; speculative trials "
                                  "of other alphas are executed but not credited",
        "rollout_steps_credited": int(st.rollout_steps), "rollout_steps_if_full": int(st.rollouts) * int(p.N),
        "frac_with_full_rollout_credit_r01": (bytes_fwd_full_rollouts / (fwd_ms * 1e-3) / 1e9 / PEAK) if fwd_ms > 0 else None,
        "timing": "hipEvents on the solver's stream around the dominant class's launches inside the timed steps; "
                  "the other classes from one untimed solve with every class bracketed (whole_solve_all_classes_ms)",
        "classes": {
            "backward(K1+K1b+K2+K3)": {"ms": bwd_ms, "GBps": gbps_bwd}, "forward(K4)": {"ms": fwd_ms, "GBps": gbps_fwd},
            "update(K4b+K5)": {"ms": upd_ms}, "whole_solve": {"ms": solve_ms, "GBps": gbps_all, "frac": gbps_all / PEAK},
            "whole_solve_all_classes_ms": float(prof.solve_ms),
        },
        "bytes_per_traj": {"fill": b_fill, "backward": b_bwd, "forward_per_alpha": b_fwd},
    }


# The other BASELINE configurations (per-GPU shares) and the BoxQP core, measured AFTER the headline loop and outside its
# timed region, so that the driver's record carries them too: (workload, solver, label)
OTHER_WORKLOADS = [
    ("cartpole", "clddp", "C2 cart-pole CLDDP (BoxQP core), B=4096"),
    ("cartpole", "logddp", "f4: cart-pole LogDDP resident on the device (relaxed log barrier of the control box), B=4096"),
    ("pendulum", "msipddp", "f4: pendulum MSIPDDP resident on the device (multiple shooting, segment length 5, control box), B=4096"),
    ("unicycle", "ipddp", "C3 unicycle N=200 box+ball, B=8192"),
    ("quadrotor", "ipddp", "C4 share: quadrotor nx=12 N=400, B=2048 (16384 / 8 GPUs)"),
    ("manip7", "ipddp", "C5 share: 7-joint arm nx=14 nu=7 N=150 terminal equality, 16 alphas, B=4096 (32768 / 8 GPUs)"),
]


def measure_other(api, workload, solver, label, steps=3, warmup=1, device=0, world=1, rank=0, dist=None, comm=None, sh=None):
    """A short measurement of one more workload with the same accounting as the headline line.

    world == 1: the workload's per-GPU share (DEFAULT_BATCH) on this GPU.  world > 1 (BASELINE configs [3] / [4], quoted on 8 GPUs):
    STRONG scaling -- the config's fixed global batch block-partitioned over the ranks, one RCCL all-gather of the 16-B records per
    step through the C-ABI, barrier + synchronize on both sides of the timed steps, MAX over ranks; every rank runs this function,
    rank 0's return value carries the line."""
    import torch
    p, spread, _ = make_problem(api, workload, solver)
    if world > 1:
        global_batch = STRONG_GLOBAL_BATCH[workload]
        lo, hi = sh.partition(global_batch, world, rank)
        cap = sh.shard_capacity(global_batch, world)
        x0 = np.ascontiguousarray(api.batch_x0(p, global_batch, 20260928 + 1, spread)[lo:hi])
    else:
        global_batch = DEFAULT_BATCH[workload]
        lo, hi, cap = 0, global_batch, global_batch
        x0 = api.batch_x0(p, global_batch, 20260928 + 1, spread)
    B = hi - lo
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B, device=device)
    gathered_dev = torch.empty(world * cap * 16, dtype=torch.uint8, device="cuda") if world > 1 else None

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        st_ = hs.solve()      # cddp_hip_solve returns after the stream is drained
        if world > 1:
            hs.allgather_results(comm, world, cap, gathered_dev.data_ptr())
        return st_
    try:
        hs.set_initial(x0, U0)
        hs.set_timing_detail(api.TIMING_ALL)
        hs.solve()      # cold first solve, not read (see main)
        prof = hs.solve()
        sweep_dominates = prof.backward_ms >= prof.forward_ms
        hs.set_timing_detail(api.TIMING_SWEEP if sweep_dominates else api.TIMING_ROLLOUT)
        for _ in range(warmup):
            step()
        sync()
        t0 = time.perf_counter()
        stats = [step() for _ in range(steps)]
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res = hs.results()
        rl = roofline_block(api, p, hs.m, B, workload, solver, stats[-1], prof, stats, sweep_dominates, concurrency=hs.concurrency(), groups=hs.num_groups())
        status_hist = {api.STATUS_STRINGS[int(s)]: int(c) for s, c in zip(*np.unique(res["status"], return_counts=True))}
        out = {
            "workload": label, "solver": solver.upper(), "batch": B, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "value": global_batch * steps / dt, "unit": "trajectories/s",
            "roofline": {"kernel": rl["kernel"], "frac": rl["frac"], "achieved": rl["achieved"], "unit": "GB/s",
                         "avg_launch_ms": rl["avg_launch_ms"], "whole_solve_frac": rl["classes"]["whole_solve"]["frac"],
                         "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"], "traffic": rl["traffic"], "traffic_source": rl["traffic_source"],
                         "concurrent_groups": rl["concurrent_groups"], "model_rate_exceeds_peak": rl["model_rate_exceeds_peak"],
                         "whole_solve_counter_based": rl["whole_solve_counter_based"]},
            "classes_ms": {k: v["ms"] for k, v in rl["classes"].items() if isinstance(v, dict)},
            "mean_iterations": float(np.mean(res["iterations"])), "status": status_hist,
        }
        if world > 1:
            rec = sh.compact_records(gathered_dev.cpu().numpy(), global_batch, world)
            assert len(rec) == global_batch
            assert np.array_equal(rec["iterations"][lo:hi], res["iterations"]) and np.array_equal(rec["status"][lo:hi], res["status"])
            out.update({"n_gpus": world, "scaling": "strong", "global_batch": global_batch, "batch_per_gpu_rank0": B,
                        "collective": "ncclAllGather, %d x %d records" % (world, cap), "rccl_ranks": api.comm_info(comm)[0],
                        "gathered_records": int(len(rec)),
                        "gathered_converged": int(np.sum((rec["status"] == api.STATUS_OPTIMAL) | (rec["status"] == api.STATUS_ACCEPTABLE)))})
            out["status"] = "rank 0's shard: %s" % status_hist
        return out
    finally:
        hs.close()


# BASELINE configs [3] and [4] are quoted on 8 GPUs: with more than one rank they are measured too (strong scaling, 2 steps each)
MULTI_RANK_WORKLOADS = [
    ("quadrotor", "ipddp", "C4: quadrotor nx=12 N=400, global batch 16384 over the ranks (BASELINE config[3])"),
    ("manip7", "ipddp", "C5: 7-joint arm nx=14 nu=7 N=150 terminal equality, 16 alphas, global batch 32768 over the ranks (BASELINE config[4])"),
]


def stackfed_bytes(nx, nu, N, m, clddp):
    """SURVEY 8(d) B_bwd per trajectory for the stack-fed sweep -- here the bytes ARE the traffic: every f_x / f_u / l_x / l_u / l_xx / l_uu / l_ux
    (+ y, s, g, G_x, G_u) row is read from HBM once, every gain / value (+ slack / dual gain) row written once."""
    dyn = nx * nx + nx * nu; cost = nx + nu + nx * nx + nu * nu + nu * nx; gain = nu * nx + nu; val = nx + nx * nx
    if clddp or m == 0:
        return 8.0 * (N * (dyn + cost + gain + val) + val)
    con_in = 3 * m + m * nx + m * nu; con_out = 2 * m + 2 * m * nx
    return 8.0 * (N * (dyn + cost + con_in + gain + val + con_out) + val)


# north_star's literal form: "coalesced HBM loads of the (N x batch) stacks of f_x / f_u / l_xx / l_uu / l_ux" -- the stack-fed sweeps of the host
# plug-in route (cddp_hip_stacks_backward, stacks.hip / stacks_coop.hpp).  (nx, nu, m, N, label, batches of the curve)
# The nx >= 12 shapes run sixteen lanes per trajectory, one 4-trajectory workgroup per SIMD: 4096 trajectories fill the chip at nx = 12
# (four workgroups per CU), 3072 at nx = 14 (three per CU: 51 KB of LDS each); the BASELINE shares (2048, 4096) are on the curve.
STACKFED_SHAPES = [
    (4, 1, 2, 100, "C2 shape", (4096, 16384, 65536, 131072)),
    (3, 2, 5, 200, "C3 shape", (8192, 32768, 65536)),
    (12, 4, 8, 400, "C4-share shape", (2048, 4096)),
    (14, 7, 14, 150, "C5-share shape", (3072, 4096)),
]
# `bench.py --stackfed`: longer curves (tens of GB of stacks for the nx >= 12 shapes: minutes of host-side tiling and upload, not part of the default line)
STACKFED_SHAPES_FULL = [
    (4, 1, 2, 100, "C2 shape", (4096, 16384, 65536, 131072, 262144)),
    (3, 2, 5, 200, "C3 shape", (8192, 32768, 65536, 131072)),
    (12, 4, 8, 400, "C4-share shape", (2048, 4096, 8192)),
    (14, 7, 14, 150, "C5-share shape", (3072, 4096, 6144, 12288)),
]


STACKFED_TRAFFIC_FILE = os.path.join(REPO, "profiles", "r06_pmc_traffic_stackfed.json")


def stackfed_traffic(nx, m, clddp, best):
    """`traffic` of a stack-fed line: FETCH_SIZE x 2 + WRITE_SIZE of ONE sweep launch of this shape / branch / batch / form (profiles/scripts/
    stackfed_traffic_r06.sh), or None where that batch was not profiled.  Every stack is read and every gain row written once -- that is the
    algorithmic figure -- and with path rows the linear-policy rollout behind the sweep re-reads f_x, f_u, K, K_s, K_y: 1.5 x."""
    try:
        tj = json.load(open(STACKFED_TRAFFIC_FILE))
    except OSError:
        return {"traffic": None, "traffic_source": "no " + os.path.relpath(STACKFED_TRAFFIC_FILE, REPO)}
    pat = re.compile(r"^nx%d_%s_(\d+)_%s_%s$" % (nx, "clddp" if clddp or m == 0 else "path", best["form"], "t4" if nx > 8 else "plain"))
    hits = sorted((abs(int(pat.match(k).group(1)) - best["batch"]), int(pat.match(k).group(1)), k) for k in tj["launches"] if pat.match(k))
    if not hits:
        return {"traffic": None, "traffic_source": "shape not in %s" % os.path.relpath(STACKFED_TRAFFIC_FILE, REPO)}
    _, pb, key = hits[0]
    byt = tj["launches"][key]["bytes_per_launch"] * best["batch"] / pb   # (every byte of a sweep is per trajectory: a launch of another batch scales)
    return {"traffic": byt, "traffic_over_algorithmic": byt / best["stack_bytes"],
            "hbm_rate_GBps": byt / best["kernel_ms"] / 1e6, "hbm_frac": byt / best["kernel_ms"] / 1e6 / 8000.0,
            "traffic_source": "%s [%s%s]: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of one launch (separate passes); with path rows the rollout behind the sweep re-reads f_x, f_u, K, K_s, K_y" % (
                os.path.relpath(STACKFED_TRAFFIC_FILE, REPO), key, "" if pb == best["batch"] else ", scaled from batch %d" % pb)}


def measure_stackfed(api, device=0, shapes=None, reps=3):
    """One line per (shape, branch): kernel time (hipEvents around the single launch, cddp_hip_stacks_last_kernel_ms, best of `reps`) of the
    stack-fed IPDDP path-row sweep and the CLDDP sweep on random well-conditioned stacks (the arithmetic does not depend on the values), the
    default sweep form of the shape, with a batch curve up to a chip-filling batch.  The stacks of the larger batches
    are the smallest batch tiled (uploaded whole: the sweep reads every row once either way)."""
    rng = np.random.default_rng(1)
    opt = api.default_options()
    PEAK = 8000.0
    lines = []
    for (nx, nu, m, N, label, batches) in (shapes or STACKFED_SHAPES):
        B0 = min(batches)
        fx = np.tile(np.eye(nx), (B0, N, 1, 1)) + 0.05 * rng.standard_normal((B0, N, nx, nx)); fu = 0.1 * rng.standard_normal((B0, N, nx, nu))
        lx = rng.standard_normal((B0, N, nx)); lu = rng.standard_normal((B0, N, nu))
        lxx = np.tile(np.eye(nx), (B0, N, 1, 1)); luu = np.tile(np.eye(nu), (B0, N, 1, 1)); lux = np.zeros((B0, N, nu, nx))
        VxN = rng.standard_normal((B0, nx)); VxxN = np.tile(10.0 * np.eye(nx), (B0, 1, 1))
        y = np.full((B0, N, m), 0.5); sl = np.full((B0, N, m), 0.4); g = -sl + 0.01 * rng.standard_normal((B0, N, m))
        Gx = 0.1 * rng.standard_normal((B0, N, m, nx)); Gu = 0.3 * rng.standard_normal((B0, N, m, nu))
        for branch, mm, bname in ((api.STACKS_IPDDP_PATH, m, "IPDDP, path rows"), (api.STACKS_CLDDP, 0, "CLDDP")):
            curve = []
            for B in batches:
                mult = (B + B0 - 1) // B0
                rep = lambda a: np.ascontiguousarray(np.concatenate([a] * mult, axis=0)[:B]) if B != B0 else a
                try:
                    hs = api.HipStackSolver(B, nx, nu, mm, N, device=device)
                except api.HipError as e:
                    curve.append({"batch": B, "error": str(e)}); continue
                try:
                    hs.set_stacks(rep(fx), rep(fu), rep(lx), rep(lu), rep(lxx), rep(luu), rep(lux), rep(VxN), rep(VxxN))
                    mu = None
                    if mm:
                        hs.set_constraint_stacks(rep(y), rep(sl), rep(g), rep(Gx), rep(Gu)); mu = np.full(B, 0.1)
                    ms = []
                    ok = None
                    for _ in range(reps + 1):
                        ok = hs.backward(branch, opt, np.full(B, 1e-6), mu, retry=False); ms.append(hs.kernel_ms())
                    t = min(ms[1:])
                    byt = stackfed_bytes(nx, nu, N, mm, branch == api.STACKS_CLDDP) * B
                    curve.append({"batch": B, "kernel_ms": t, "GBps": byt / t / 1e6, "frac": byt / t / 1e6 / PEAK, "form": "coop" if hs.sweep_form() == 1 else "lane",
                                  "sweeps_ok": int(ok.sum()), "stack_bytes": byt})
                finally:
                    hs.close()
            good = [c for c in curve if "frac" in c]
            best = max(good, key=lambda c: c["frac"]) if good else None
            lines.append({
                "workload": "stack-fed sweep (g1, cddp_hip_stacks_backward): %s nx=%d nu=%d m=%d N=%d, %s" % (label, nx, nu, mm, N, bname),
                "solver": "stack-fed " + bname, "unit": "trajectory sweeps/s",
                "batch": best["batch"] if best else None, "value": (best["batch"] / (best["kernel_ms"] * 1e-3)) if best else None,
                "ms_per_step": best["kernel_ms"] if best else None,
                "roofline": None if not best else {"bound": "hbm", "kernel": "k_stacks_backward" + ("_coop" if best["form"] == "coop" else ""), "achieved": best["GBps"], "peak": PEAK,
                                                   "unit": "GB/s", "frac": best["frac"], "algorithmic_bytes_per_launch": best["stack_bytes"],
                                                   **stackfed_traffic(nx, mm, branch == api.STACKS_CLDDP, best)},
                "batch_curve": curve,
            })
    return lines


def measure_plugin(api, device=0, batch=4096):
    """One cddp_hip_plugin_solve line: the control-limited pendulum written as a USER plug-in in C (tests/cpp/pendulum_plugin.c:
    dynamics, Jacobians, cost and constraint callbacks), IPDDP, the batch's host work on every CPU the lease owns (cddp_hip_plugin_set_host_threads),
    the backward passes as stack-fed GPU sweeps.  Reported: trajectories / s and where the time goes (host callbacks + line search vs GPU sections)."""
    import ctypes as C
    so = "/tmp/cddp_pendulum_plugin_%d.so" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(REPO, "tests", "cpp", "pendulum_plugin.c"), "-lm"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    pl = C.CDLL(so)

    class Params(C.Structure):
        _fields_ = [(n, C.c_double) for n in ("dt", "length", "mass", "damping", "gravity", "Qf", "R", "umax")]
    prm = Params(0.02, 0.5, 1.0, 0.01, 9.81, 100.0, 0.1, 20.0)
    p = api.pendulum_problem(api.SOLVER_IPDDP, True)
    lib = api.load_hip()
    ps = api.PluginStruct()
    ps.abi_version = api.ABI_VERSION; ps.options_bytes = C.sizeof(api.Options); ps.user = C.cast(C.pointer(prm), C.c_void_p)
    ps.nx, ps.nu, ps.n_constraints = 2, 1, 1
    ps.constraint_dims[0] = 2
    for field, sym, ftype in (("discrete_dynamics", "pend_dynamics", api._F_DYN), ("jacobians", "pend_jacobians", api._F_JAC), ("running_cost", "pend_running_cost", api._F_RC),
                              ("terminal_cost", "pend_terminal_cost", api._F_TC), ("running_cost_derivatives", "pend_running_cost_derivatives", api._F_RCD),
                              ("terminal_cost_derivatives", "pend_terminal_cost_derivatives", api._F_TCD), ("constraints", "pend_constraints", api._F_CON)):
        setattr(ps, field, C.cast(getattr(pl, sym), ftype))
    B, N = batch, 100
    x0 = api.batch_x0(p, B, 20260928 + 7, [0.1, 0.1])
    res = np.zeros(B, dtype=api.RESULT_DTYPE)
    ncpu, _ = available_cpus()
    out = {}
    for threads in (1, ncpu):
        lib.cddp_hip_plugin_set_host_threads(int(threads))
        t0 = time.perf_counter()
        rc = lib.cddp_hip_plugin_solve(C.byref(ps), int(api.SOLVER_IPDDP), N, C.c_double(0.02), C.byref(p.options), int(device), B, api._ptr(x0), None, None,
                                       res.ctypes.data_as(C.c_void_p), None, None, None)
        wall = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("cddp_hip_plugin_solve: %s" % lib.cddp_hip_last_error().decode())
        tot, gpu, ker = C.c_double(), C.c_double(), C.c_double(); sw, th = C.c_int(), C.c_int()
        lib.cddp_hip_plugin_last_stats(C.byref(tot), C.byref(gpu), C.byref(ker), C.byref(sw), C.byref(th))
        out[threads] = {"threads": th.value, "wall_ms": wall * 1e3, "value": B / wall, "gpu_section_ms": gpu.value, "sweep_kernel_ms": ker.value, "batch_sweeps": sw.value,
                        "host_ms": tot.value - gpu.value, "host_fraction": (tot.value - gpu.value) / max(tot.value, 1e-9)}
    lib.cddp_hip_plugin_set_host_threads(1)
    best = out[ncpu]
    conv = int(np.sum((res["status"] == api.STATUS_OPTIMAL) | (res["status"] == api.STATUS_ACCEPTABLE)))
    byt = stackfed_bytes(2, 1, N, 2, False) * B * best["batch_sweeps"]
    return {"workload": "host plug-in solve (g1, cddp_hip_plugin_solve): pendulum nx=2 nu=1 N=100 control box as C callbacks (tests/cpp/pendulum_plugin.c), IPDDP, B=%d" % B,
            "solver": "IPDDP (plug-in route)", "batch": B, "unit": "trajectories/s", "value": best["value"], "ms_per_step": best["wall_ms"],
            "host_threads": best["threads"], "time_split": best, "single_thread": out[1], "mean_iterations": float(np.mean(res["iterations"])), "converged": conv,
            "roofline": {"bound": "hbm", "kernel": "k_stacks_backward (inside the plug-in solve)", "achieved": byt / (best["sweep_kernel_ms"] * 1e-3) / 1e9 if best["sweep_kernel_ms"] > 0 else None,
                         "peak": 8000.0, "unit": "GB/s", "frac": (byt / (best["sweep_kernel_ms"] * 1e-3) / 1e9 / 8000.0) if best["sweep_kernel_ms"] > 0 else None,
                         "note": "the sweeps' kernel time only; the call is bound by the host side (time_split.host_fraction)"}}


def measure_mpc(api, rounds, device=0, workload="cartpole", batch=0):
    """f1 caller side: receding-horizon re-solves on a RE-USED handle -- the caller pattern of
    examples/ipddp_mpcc_rc.py:649-705 (solver.set_initial_state(state); solver.solve()) with the reference's warm-start branch
    "existing solver state" (ipddp_solver.cpp:675-731: controls, slacks, duals, gains and regularisation kept).  Round 0 is the cold
    solve; round k >= 1 starts every trajectory from the state its own previous plan predicted one step ahead (x_1 of the previous
    solution: the plant is the model) and re-solves.  Few iterations per solve: launches and host polls, not kernels, dominate --
    the line reports the device time (hipEvents, begin .. end of cddp_hip_solve's stream work) next to the wall time of the call."""
    p, spread, desc = make_problem(api, workload, "ipddp")
    B = batch if batch > 0 else DEFAULT_BATCH[workload]
    x0 = api.batch_x0(p, B, 20260928 + 1, spread)
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B, device=device)
    try:
        hs.set_initial(x0, U0)
        hs.solve()                      # code load / first touch
        hs.set_initial(x0, U0)
        t0 = time.perf_counter(); st0 = hs.solve(); cold_wall = time.perf_counter() - t0
        r0 = hs.results()
        hs.set_warm_start(True)
        wall_solve = []; wall_step = []; dev_ms = []; iters = []; launches = []; conv = []
        for _ in range(rounds):
            t_step = time.perf_counter()
            u0, x1 = hs.plan_head()     # the head of the plan goes back to the host: an MPC loop applies u_0 and restarts from x_1
            hs.set_initial_state(x1)
            t1 = time.perf_counter(); st = hs.solve(); t2 = time.perf_counter()
            r = hs.results()
            wall_solve.append(t2 - t1); wall_step.append(t2 - t_step); dev_ms.append(st.solve_ms); launches.append(int(st.kernel_launches))
            iters.append(float(np.mean(r["iterations"])))
            conv.append(int(np.sum((r["status"] == api.STATUS_OPTIMAL) | (r["status"] == api.STATUS_ACCEPTABLE))))
        ws = float(np.mean(wall_solve)); wd = float(np.mean(dev_ms)) * 1e-3
        # ---- the reference's OTHER warm start: a caller that builds a fresh problem per MPC step and seeds it with the
        # previous plan shifted by one step (X_[1:] + repeated last state, U_[1:] + repeated last control) takes the "provided trajectory" branch
        # (ipddp_solver.cpp:733-816: barrier parameter from the seed's largest violation, duals re-initialised, gains zero).  Same handle,
        # cddp_hip_forget_solver_state between steps; the cold solve's plan is the first seed.
        prov = None
        try:
            hs.set_warm_start(False); hs.set_initial(x0, U0); hs.solve(); hs.set_warm_start(True)
            p_wall = []; p_dev = []; p_it = []; p_conv = []
            for _ in range(rounds):
                X, U = hs.trajectory()
                Xs = np.concatenate([X[:, 1:], X[:, -1:]], axis=1); Us = np.concatenate([U[:, 1:], U[:, -1:]], axis=1)
                hs.forget_solver_state()
                hs.set_initial(np.ascontiguousarray(Xs[:, 0]), np.ascontiguousarray(Us), np.ascontiguousarray(Xs))
                t1 = time.perf_counter(); st = hs.solve(); t2 = time.perf_counter()
                r = hs.results()
                p_wall.append(t2 - t1); p_dev.append(st.solve_ms); p_it.append(float(np.mean(r["iterations"])))
                p_conv.append(int(np.sum((r["status"] == api.STATUS_OPTIMAL) | (r["status"] == api.STATUS_ACCEPTABLE))))
            prov = {"warm_start": "provided trajectory (previous plan shifted by one step) on a forgotten solver state: cddp_hip_forget_solver_state + cddp_hip_set_initial + cddp_hip_solve",
                    "value": B / float(np.mean(p_wall)), "ms_per_step": float(np.mean(p_wall)) * 1e3, "device_ms_per_resolve": float(np.mean(p_dev)),
                    "mean_iterations_per_resolve": float(np.mean(p_it)), "iterations_by_round": p_it, "converged_by_round": p_conv,
                    "note": "the timed call excludes the host-side shift and the 2 x upload of the seed (set_initial)"}
        except Exception as e:   # the second mode must not cost the first its line
            prov = {"error": repr(e)}
        return {
            "provided_trajectory_warm_start": prov,
            "workload": "MPC re-solves (f1): %s, B=%d, %d shift-by-one-step re-solves on a re-used handle, warm start = existing solver state" % (desc, B, rounds),
            "solver": "IPDDP", "batch": B, "steps": rounds, "unit": "re-solved trajectories/s",
            "value": B / ws, "ms_per_step": ws * 1e3, "ms_per_mpc_step_incl_readback": float(np.mean(wall_step)) * 1e3,
            "device_ms_per_resolve": wd * 1e3, "host_fraction_of_solve_call": max(0.0, 1.0 - wd / ws),
            "mean_iterations_per_resolve": float(np.mean(iters)), "iterations_by_round": iters, "kernel_launches_per_resolve": float(np.mean(launches)),
            "converged_by_round": conv,
            "cold": {"ms": cold_wall * 1e3, "device_ms": float(st0.solve_ms), "mean_iterations": float(np.mean(r0["iterations"])),
                     "converged": int(np.sum((r0["status"] == api.STATUS_OPTIMAL) | (r0["status"] == api.STATUS_ACCEPTABLE)))},
        }
    finally:
        hs.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: the BASELINE config's per-GPU batch of the workload)")
    ap.add_argument("--solver", default="ipddp", choices=["ipddp", "clddp", "logddp", "msipddp"])
    ap.add_argument("--workload", default="cartpole", choices=["cartpole", "cartpole_unc", "unicycle", "pendulum", "quadrotor", "manip7"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short C2-CLDDP / C3 / C4-share / C5-share measurements appended to the default (C2-IPDDP, 1 GPU) line")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch trajectories per GPU; strong: a fixed global batch (--global-batch or the BASELINE config's) over all GPUs")
    ap.add_argument("--global-batch", type=int, default=0)
    ap.add_argument("--stackfed", action="store_true", help="only the stack-fed sweep lines (batch curves), one JSON line")
    ap.add_argument("--mpc", type=int, default=0, help="only the MPC re-solve measurement: K shift-and-re-solve rounds on a re-used handle (1 GPU), one JSON line")
    args = ap.parse_args()

    import torch
    launched_by_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the cddp_hip solver core has no CPU fallback")
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to report an N-GPU line measured on fewer devices"
                         % (args.gpus, torch.cuda.device_count()))
    if args.gpus > 1 and not launched_by_torchrun:
        # Plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, RCCL over xGMI) instead of
        # silently measuring one rank.  The children see RANK / WORLD_SIZE and take the normal path; rank 0 prints the line.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d: launch with --nproc-per-node %d (or without torchrun: bench.py starts its own ranks)"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or launched_by_torchrun:   # (also with one rank under torchrun: the RCCL path then runs with a size-1 communicator)
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    api = load_api()
    if args.stackfed:
        print(json.dumps({"stackfed": measure_stackfed(api, device=local_rank, shapes=STACKFED_SHAPES_FULL), "plugin": measure_plugin(api, device=local_rank)}))
        return
    if args.mpc > 0:
        if world != 1:
            raise SystemExit("--mpc is a single-GPU measurement")
        print(json.dumps(measure_mpc(api, args.mpc, device=local_rank, workload=args.workload, batch=args.batch)))
        return
    p, spread, desc = make_problem(api, args.workload, args.solver)
    sh = load_module("cddp_sharding", "sharding.py")
    # one seeded global batch (seed = 20260928 + config_index), block-partitioned over the ranks.
    #   weak   (default): B trajectories per GPU, global batch B * world;
    #   strong          : the config's fixed global batch (BASELINE configs 4 / 5: 16384 / 32768) cut into `world` blocks,
    #                     uneven blocks padded in the gather (sharding.shard_capacity).
    if args.scaling == "strong":
        global_batch = args.global_batch if args.global_batch > 0 else STRONG_GLOBAL_BATCH[args.workload]
    else:
        global_batch = (args.batch if args.batch > 0 else DEFAULT_BATCH[args.workload]) * world
    lo, hi = sh.partition(global_batch, world, rank)
    B = hi - lo
    cap = sh.shard_capacity(global_batch, world)
    x0_global = api.batch_x0(p, global_batch, 20260928 + 1, spread)
    x0 = np.ascontiguousarray(x0_global[lo:hi])
    del x0_global
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B, device=local_rank)
    hs.set_initial(x0, U0)                     # H2D once; solve() restarts from the device-resident copy
    # The path's single collective goes through the C-ABI (cddp_hip_allgather_results = ncclAllGather on a communicator
    # made with cddp_hip_comm_init); torch.distributed only carries the 128-byte unique id, the barrier and the timing max.
    comm = None
    if dist is not None:
        idt = torch.zeros(api.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        comm = api.comm_init(bytes(idt.cpu().numpy().tobytes()), world, rank, local_rank)
    gathered_dev = torch.empty(world * cap * 16, dtype=torch.uint8, device="cuda")

    def step():
        st = hs.solve()
        hs.allgather_results(comm, world, cap, gathered_dev.data_ptr())   # the single RCCL collective of the path
        return st, gathered_dev

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # One untimed solve with every kernel class bracketed by hipEvents (the "classes" block below and the choice of
    # the dominant class); the timed steps then bracket ONLY the dominant class's launches -- every event costs ~5 us
    # of queue time, 6 per iteration are ~5 % of a C2 solve.
    if comm is not None:   # connection set-up (lazy in RCCL) is not part of a step, whatever --warmup says
        hs.allgather_results(comm, world, cap, gathered_dev.data_ptr())
    hs.set_timing_detail(api.TIMING_ALL)
    hs.solve()          # the handle's first solve (code load, first touch of the stacks, clock ramp) is not the one that is read
    prof = hs.solve()
    sweep_dominates = prof.backward_ms >= prof.forward_ms
    hs.set_timing_detail(api.TIMING_SWEEP if sweep_dominates else api.TIMING_ROLLOUT)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    stats = []
    gathered = None
    for _ in range(args.steps):
        st_, gathered = step()
        stats.append(st_)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    # work counters of this rank's last step (identical work every step: same inputs)
    st = stats[-1]
    res = hs.results()
    m = hs.m
    ipddp = args.solver == "ipddp"
    solve_ms = float(np.mean([s.solve_ms for s in stats]))
    roofline = roofline_block(api, p, m, B, args.workload, args.solver, st, prof, stats, sweep_dominates, concurrency=hs.concurrency(), groups=hs.num_groups())
    rec = sh.compact_records(gathered.cpu().numpy(), global_batch, world)   # drops (and checks) the padding of uneven shards
    assert len(rec) == global_batch
    assert np.array_equal(rec["iterations"][lo:hi], res["iterations"]) and np.array_equal(rec["status"][lo:hi], res["status"])
    gathered_converged = int(np.sum((rec["status"] == api.STATUS_OPTIMAL) | (rec["status"] == api.STATUS_ACCEPTABLE)))
    status_hist = {api.STATUS_STRINGS[int(s)]: int(c) for s, c in zip(*np.unique(res["status"], return_counts=True))}
    total_traj = global_batch * args.steps
    out = {
        "metric": "trajectories_per_sec", "value": total_traj / dt_max, "unit": "trajectories/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "parity": parity_block(),
        "config": {
            "workload": {"cartpole": "BASELINE config[1]: ", "unicycle": "BASELINE config[2]: ", "quadrotor": "BASELINE config[3] (one GPU share): ", "manip7": "BASELINE config[4] (one GPU share): "}.get(args.workload, "experiment: ") + desc +
                        ", batch %d per GPU, solver %s" % (B, args.solver.upper()),
            "solver": args.solver.upper(), "batch_per_gpu": B, "global_batch": global_batch, "nx": p.nx, "nu": p.nu,
            "horizon": p.N, "path_dual_dim": m, "max_iterations": int(p.options.max_iterations),
            "line_search": "%s rule, %d alphas" % ("best-merit (enable_parallel)" if p.options.enable_parallel else "first-success", int(p.options.ls_max_iterations)),
            "sharding": "independent trajectories, block partition (%s scaling), one RCCL all-gather of 16-B records per step through "
                        "the C-ABI (cddp_hip_allgather_results)" % args.scaling,
            "collective": "ncclAllGather, %d x %d records" % (world, cap) if comm is not None else "none (1 GPU, device copy)",
        },
        "solve": {
            "mean_iterations": float(np.mean(res["iterations"])), "max_iterations": int(np.max(res["iterations"])),
            "status": status_hist, "sweeps": int(st.sweeps), "rollouts_useful": int(st.rollouts),
            "rollouts_launched": int(st.rollouts_launched), "kernel_launches": int(st.kernel_launches),
            "device_solve_ms": solve_ms, "gathered_records": int(len(rec)), "gathered_converged": gathered_converged,
        },
        "roofline": roofline,
    }
    assert out["n_gpus"] == args.gpus
    if comm is not None:
        n_rccl, r_rccl = api.comm_info(comm)     # what RCCL itself says, not this script's bookkeeping
        assert n_rccl == world and r_rccl == rank, (n_rccl, world, r_rccl, rank)
        out["config"]["rccl_ranks"] = n_rccl
    hs.close()
    headline = args.workload == "cartpole" and args.solver == "ipddp" and args.batch in (0, 4096) and args.scaling == "weak"
    if world == 1 and dist is None and not args.no_other_workloads and headline:
        # the other BASELINE configurations, 3 steps each, after and outside the headline's timed region
        out["other_workloads"] = []
        for wl, sv, label in OTHER_WORKLOADS:
            try:
                out["other_workloads"].append(measure_other(api, wl, sv, label, device=local_rank))
            except Exception as e:   # a failing extra workload must not lose the headline line
                out["other_workloads"].append({"workload": label, "error": "%s: %s" % (type(e).__name__, e)})
        # pendulum and unicycle (C3): converging solves, where a warm start pays (few iterations per re-solve); the cart-pole example never
        # converges inside its 80-iteration cap (cold or warm), so its re-solve line would measure the cap, not warm starts: dropped
        try:   # north_star's literal form: the stack-fed sweeps of the plug-in route
            out["other_workloads"].extend(measure_stackfed(api, device=local_rank))
        except Exception as e:
            out["other_workloads"].append({"workload": "stack-fed sweeps", "error": "%s: %s" % (type(e).__name__, e)})
        try:
            out["other_workloads"].append(measure_plugin(api, device=local_rank))
        except Exception as e:
            out["other_workloads"].append({"workload": "host plug-in solve", "error": "%s: %s" % (type(e).__name__, e)})
        for wl in ("pendulum", "unicycle"):
            try:
                out["other_workloads"].append(measure_mpc(api, 8, device=local_rank, workload=wl))
            except Exception as e:
                out["other_workloads"].append({"workload": "MPC re-solves (%s)" % wl, "error": "%s: %s" % (type(e).__name__, e)})
    elif dist is not None and not args.no_other_workloads and headline:
        # under torchrun (any world size, 1 included): configs [3] / [4] with the batch partitioned over the ranks and the same
        # single collective -- every rank takes part, rank 0 keeps the lines
        out["other_workloads"] = []
        for wl, sv, label in MULTI_RANK_WORKLOADS:
            try:
                if world > 1:
                    r = measure_other(api, wl, sv, label, steps=2, warmup=0, device=local_rank, world=world, rank=rank, dist=dist, comm=comm, sh=sh)
                else:   # one rank under torchrun: the per-GPU share, size-1 communicator already exercised by the headline
                    r = measure_other(api, wl, sv, label + " -- one rank: the 8-GPU share", steps=2, warmup=0, device=local_rank, dist=dist)
                    r.update({"n_gpus": 1, "rccl_ranks": api.comm_info(comm)[0]})
                out["other_workloads"].append(r)
            except Exception as e:
                if world > 1:
                    raise          # a rank that drops out would hang the others in the next barrier: fail the job loudly
                out["other_workloads"].append({"workload": label, "error": "%s: %s" % (type(e).__name__, e)})
    if comm is not None:
        api.comm_destroy(comm)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(api, p, x0, U0)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
