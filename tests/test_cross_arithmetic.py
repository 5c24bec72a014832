"""Cross-arithmetic yardstick.

Since round 4 every strict `-m gpu` comparison runs the checker in the library's OWN sin / cos / log / pow (oracle trig_mode 1 =
cddp-cpp_amd/csrc/dev_trig.hpp): strict, but common-mode in those routines.  The reference itself evaluates them with glibc.  This
module keeps the other yardstick beside the strict one: the five benchmarked batches (bench.py's own inputs) and the two resident f4
batches solved by the shipped library and by the checker in GLIBC arithmetic (trig_mode 0), the fraction of trajectories whose
decisions differ REPORTED into gpurun_out/parity_report_cross_*.json (copied to profiles/ and carried in the bench line's `parity`
block) and bounded -- not asserted strict: two correct solvers whose sines differ in the last bit legitimately take different
line-search decisions on knife-edge plants (tests/golden/trig_noise_flip_rates.json measures how often 1-ulp noise inside the
CHECKER alone flips them; the library may not flip more often than that noise does)."""
import json
import os

import numpy as np
import pytest

from test_full_size import BENCH_SEED, _bench_problem, _report, _solve

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trig_noise_flip_rates.json")) as _f:
    TRIG_NOISE = json.load(_f)

# workload -> (trajectories compared (0 = the whole batch), bound on the (status, iterations) mismatch fraction, yardstick entry or None)
#   whole batches: the plants whose accept / reject path showed no knife edge in rounds 2 - 3 (device libm vs glibc: 0 of 4096 / 4096 /
#   8192 differed, profiles/r02_parity_report.md) -- bound 1 %;
#   C4 / C5 shares: 32 trajectories against the 1-ulp-noise yardstick of the same plant (same_counts of 32, margin 3 as in round 3).
CASES = {
    "cartpole": (0, 0.01, None), "cartpole_clddp": (0, 0.01, None), "unicycle": (0, 0.01, None),
    "cartpole_logddp": (0, 0.02, None), "pendulum_msipddp": (0, 0.02, None),
    "quadrotor": (32, None, "quad12_ipddp_box"), "manip7": (32, None, "manip7_term_eq_parallel_ls"),
}
KNIFE_MARGIN = 3


@pytest.mark.parametrize("workload", list(CASES))
def test_bench_batch_against_glibc_checker(api, oracle_built, workload):
    n_cmp, bound, yard = CASES[workload]
    p, spread, B = _bench_problem(api, workload)
    x0 = api.batch_x0(p, B, BENCH_SEED, spread)
    U0 = api.batch_U0(p, B)
    r, X, U, K, k = _solve(api, p, x0, U0)
    idx = np.arange(B) if n_cmp == 0 else np.unique(np.concatenate([[0, 1, 63, 64, B // 2, B - 1], np.arange(200, 200 + n_cmp)]))[:n_cmp]
    prev = api.set_trig_mode(0)          # the reference's arithmetic: glibc sin / cos / log / pow
    try:
        ores, _, _, _, ms = api.oracle_solve_batch(p, np.ascontiguousarray(x0[idx]), None if U0 is None else np.ascontiguousarray(U0[idx]),
                                                   n_threads=os.cpu_count() or 8, want_traj=False)
    finally:
        api.set_trig_mode(prev)
    same_counts = (r["iterations"][idx] == ores["iterations"]) & (r["status"][idx] == ores["status"])
    same_work = same_counts & (r["n_backward"][idx] == ores["n_backward"]) & (r["n_forward"][idx] == ores["n_forward"])
    rel = np.abs(r["final_objective"][idx] - ores["final_objective"]) / np.maximum(1.0, np.abs(ores["final_objective"]))
    conv_o = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    conv_h = (r["status"][idx] == api.STATUS_OPTIMAL) | (r["status"][idx] == api.STATUS_ACCEPTABLE)
    both = conv_o & conv_h
    rep = {"workload": workload, "checker_arithmetic": "glibc (oracle trig_mode 0)", "library_arithmetic": "shared straight-line (dev_trig.hpp)",
           "compared": int(len(idx)), "batch": int(B),
           "count_flip_frac": float(1.0 - same_counts.mean()), "work_flip_frac": float(1.0 - same_work.mean()),
           "objective_1e-7_mismatch_frac": float(np.mean(rel > 1e-7)), "objective_1e-4_mismatch_frac_both_converged": float(np.mean(rel[both] > 1e-4)) if both.any() else 0.0,
           "converged_checker": int(conv_o.sum()), "converged_library": int(conv_h.sum()),
           "mean_iterations_checker": float(np.mean(ores["iterations"])), "mean_iterations_library": float(np.mean(r["iterations"][idx])),
           "yardstick": None if yard is None else {"case": yard, "same_counts_of_32_under_1ulp_noise": TRIG_NOISE[yard]["same_counts"]}}
    _report("cross_" + workload, rep)
    if bound is not None:
        assert rep["count_flip_frac"] <= bound, rep
    else:
        assert same_counts.sum() >= TRIG_NOISE[yard]["same_counts"] * len(idx) / 32.0 - KNIFE_MARGIN, rep
    # where both sides converge they converge to the same optimum, whatever path the last bits chose
    assert rep["objective_1e-4_mismatch_frac_both_converged"] <= 0.01, rep
    # converged counts agree within the flip bound
    assert abs(int(conv_o.sum()) - int(conv_h.sum())) <= max(3, int(0.01 * len(idx))), rep


# ---- Round 6: summation ORDER (VERDICT r05 weak #1c / next-round 4).  The kernels and the checker accumulate every sum serially; Eigen 3.4
# in the reference's plain x86-64 build uses 2-wide SSE2 packets for dot products, norms and transposed matrix-vector products, i.e. another
# association for every sum of four or more terms (oracle/linalg.hpp::assoc_mode restates that order from Eigen's kernels -- a reading, not a
# measurement: no Eigen in this image).  Here the shipped library is held against the checker in the SAME elementary functions (trig_mode 1:
# the comparison isolates the order) but in Eigen's association; flip rates are reported like the glibc ones and bounded by the same rules.
@pytest.mark.parametrize("workload", list(CASES))
def test_bench_batch_against_eigen_order_checker(api, oracle_built, workload):
    n_cmp, bound, yard = CASES[workload]
    p, spread, B = _bench_problem(api, workload)
    x0 = api.batch_x0(p, B, BENCH_SEED, spread)
    U0 = api.batch_U0(p, B)
    r, X, U, K, k = _solve(api, p, x0, U0)
    idx = np.arange(B) if n_cmp == 0 else np.unique(np.concatenate([[0, 1, 63, 64, B // 2, B - 1], np.arange(200, 200 + n_cmp)]))[:n_cmp]
    api.set_assoc_mode(1)
    try:
        ores, _, _, _, ms = api.oracle_solve_batch(p, np.ascontiguousarray(x0[idx]), None if U0 is None else np.ascontiguousarray(U0[idx]),
                                                   n_threads=os.cpu_count() or 8, want_traj=False)
    finally:
        api.set_assoc_mode(0)
    same_counts = (r["iterations"][idx] == ores["iterations"]) & (r["status"][idx] == ores["status"])
    same_work = same_counts & (r["n_backward"][idx] == ores["n_backward"]) & (r["n_forward"][idx] == ores["n_forward"])
    rel = np.abs(r["final_objective"][idx] - ores["final_objective"]) / np.maximum(1.0, np.abs(ores["final_objective"]))
    conv_o = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    conv_h = (r["status"][idx] == api.STATUS_OPTIMAL) | (r["status"][idx] == api.STATUS_ACCEPTABLE)
    both = conv_o & conv_h
    rep = {"workload": workload, "checker_summation": "Eigen 3.4 SSE2 packet order as restated in oracle/linalg.hpp (assoc_mode 1)", "library_summation": "serial, left to right",
           "elementary_functions": "shared straight-line routines on both sides (trig_mode 1)",
           "compared": int(len(idx)), "batch": int(B), "nx": int(p.nx), "nu": int(p.nu),
           "bitwise_equal_objective_frac": float(np.mean(r["final_objective"][idx] == ores["final_objective"])),
           "count_flip_frac": float(1.0 - same_counts.mean()), "work_flip_frac": float(1.0 - same_work.mean()),
           "objective_1e-7_mismatch_frac": float(np.mean(rel > 1e-7)), "objective_1e-4_mismatch_frac_both_converged": float(np.mean(rel[both] > 1e-4)) if both.any() else 0.0,
           "converged_checker": int(conv_o.sum()), "converged_library": int(conv_h.sum()),
           "mean_iterations_checker": float(np.mean(ores["iterations"])), "mean_iterations_library": float(np.mean(r["iterations"][idx])),
           "yardstick": None if yard is None else {"case": yard, "same_counts_of_32_under_1ulp_matmul_noise": TRIG_NOISE[yard]["matmul_noise_same_counts"]}}
    _report("assoc_" + workload, rep)
    if bound is not None:
        assert rep["count_flip_frac"] <= max(bound, 0.02), rep
    else:
        assert same_counts.sum() >= TRIG_NOISE[yard]["matmul_noise_same_counts"] * len(idx) / 32.0 - KNIFE_MARGIN, rep
    assert rep["objective_1e-4_mismatch_frac_both_converged"] <= 0.01, rep
