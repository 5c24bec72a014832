"""Round-2 parity hardening (VERDICT r01 "next round" item 1): the cases the round-1 suite waived.

  * BIG_CASES (nu >= 2 BoxQP, quadrotor, manipulator) at SOLVE level, not only the first sweep;
  * the best-merit (enable_parallel) line-search rule on plants WITHOUT the knife-edge waiver;
  * heun / rk3 integrators;
  * gains / value function at LATE iterates (small mu, Y S^-1 near the 1e6 clip), 1e-8;
  * knife-edge plants (sin / cos in the dynamics AND binding fraction-to-boundary caps): the decision-flip rate is
    MEASURED over a batch of 32 and bounded, instead of "half of four trajectories agree".

Every comparison is HIP (through the C-ABI) vs the CPU oracle on the same seeded inputs.
"""
import json
import os

import numpy as np
import pytest

from test_gpu_parity import BIG_CASES, KNIFE_EDGE_CASES, TERM_CASES, TOL, make, rel_err, spread_for

pytestmark = pytest.mark.gpu

REPORT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _report(name, obj):
    """Measured agreement figures go to gpurun_out/ (merged back from the GPU box) as well as to the pytest log."""
    try:
        os.makedirs(REPORT_DIR, exist_ok=True)
        with open(os.path.join(REPORT_DIR, "parity_report_%s.json" % name), "w") as f:
            json.dump(obj, f)
    except OSError:
        pass
    print("[parity-report] %s %s" % (name, json.dumps(obj)))


def _inputs(api, p, B, seed):
    x0 = api.batch_x0(p, B, seed, spread_for(p))
    U0 = api.batch_U0(p, B)
    X0 = np.tile(p.X0_single, (B, 1, 1)) if hasattr(p, "X0_single") else None
    if X0 is not None:
        X0[:, 0, :] = x0
    return x0, U0, X0


def _solve_both(api, p, B, seed, threads=8):
    x0, U0, X0 = _inputs(api, p, B, seed)
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0, U0, X0)
    hs.solve()
    res = hs.results(); X, U = hs.trajectory(); K, k = hs.gains()
    hs.close()
    ores, oX, oU, oK, _ = api.oracle_solve_batch(p, x0, U0, X0, n_threads=threads)
    return res, X, U, K, ores, oX, oU, oK


def _agreement(api, res, ores, X, oX, U, oU, K, oK):
    B = len(res)
    same_counts = (res["iterations"] == ores["iterations"]) & (res["status"] == ores["status"])
    same_work = same_counts & (res["n_backward"] == ores["n_backward"]) & (res["n_forward"] == ores["n_forward"])
    strict = np.zeros(B, dtype=bool)
    for b in range(B):
        strict[b] = bool(same_work[b] and rel_err(res["final_objective"][b], ores["final_objective"][b]) < 1e-7
                         and rel_err(X[b], oX[b]) < 1e-6 and rel_err(U[b], oU[b]) < 1e-6 and rel_err(K[b], oK[b]) < 1e-4)
        # (gains of the FINAL iterate after dozens of iterations: X, U agree to ~1e-9 where K, through the factorisation of an
        #  ill-conditioned Q_uu, shows 6e-5 on one 68-iteration unicycle/heun solve; the 1e-8 gain bar is the sweep-level one,
        #  test_late_iterate_gains / test_step_level_parity)
    conv = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    return same_counts, same_work, strict, conv


# ------------------------------------------------------------------------------------------------------------------
# (c) BIG_CASES at solve level.  unicycle / quadrotor / manipulator CLDDP run the nu >= 2 BoxQP across iterations.
# ------------------------------------------------------------------------------------------------------------------
STRICT_BIG = ["unicycle_clddp_box", "quadrotor_clddp_box", "manipulator_clddp_box"]


@pytest.mark.parametrize("case", STRICT_BIG)
def test_big_cases_full_solve_strict(api, oracle_built, case):
    """CLDDP has no fraction-to-boundary rule, so none of these is a knife-edge case: every trajectory must agree in
    status / iteration count; those that converge also in sweep / rollout counts, objective 1e-7, X, U 1e-6, K 1e-5."""
    p = make(api, case)
    B = 8
    res, X, U, K, ores, oX, oU, oK = _solve_both(api, p, B, 20260929)
    same_counts, same_work, strict, conv = _agreement(api, res, ores, X, oX, U, oU, K, oK)
    _report("big_" + case, {"B": B, "same_counts": int(same_counts.sum()), "strict": int(strict.sum()), "converged": int(conv.sum())})
    assert same_counts.all(), (case, list(zip(res["iterations"], ores["iterations"], res["status"], ores["status"])))
    assert strict[conv].all(), (case, strict, conv)
    assert strict[0]


# ------------------------------------------------------------------------------------------------------------------
# (c) best-merit rule (cddp_solver_base.cpp:264-314) outside the knife-edge waiver; heun / rk3 integrators
# ------------------------------------------------------------------------------------------------------------------
def _parallel(api, name):
    p = make(api, name)
    p.options.enable_parallel = 1
    return p


def _integrator(api, name, integ):
    p = make(api, name)
    p.c.integrator = integ
    return p


VARIANTS = {
    "pendulum_ipddp_box_parallel": lambda S: _parallel(S, "pendulum_ipddp_box"),
    "pendulum_clddp_box_parallel": lambda S: _parallel(S, "pendulum_clddp_box"),
    "cartpole_ipddp_box_parallel": lambda S: _parallel(S, "cartpole_ipddp_box"),
    "cartpole_clddp_box_parallel": lambda S: _parallel(S, "cartpole_clddp_box"),
    "unicycle_ipddp_box_ball_parallel": lambda S: _parallel(S, "unicycle_ipddp_box_ball"),
    "unicycle_clddp_box_parallel": lambda S: _parallel(S, "unicycle_clddp_box"),
    "pendulum_ipddp_box_heun": lambda S: _integrator(S, "pendulum_ipddp_box", S.HEUN),
    "pendulum_clddp_box_rk3": lambda S: _integrator(S, "pendulum_clddp_box", S.RK3),
    "cartpole_ipddp_box_heun": lambda S: _integrator(S, "cartpole_ipddp_box", S.HEUN),
    "cartpole_ipddp_box_rk3": lambda S: _integrator(S, "cartpole_ipddp_box", S.RK3),
    "cartpole_clddp_box_heun": lambda S: _integrator(S, "cartpole_clddp_box", S.HEUN),
    "unicycle_ipddp_box_ball_rk3": lambda S: _integrator(S, "unicycle_ipddp_box_ball", S.RK3),
    "unicycle_ipddp_box_ball_heun": lambda S: _integrator(S, "unicycle_ipddp_box_ball", S.HEUN),
}


@pytest.mark.parametrize("case", list(VARIANTS))
def test_variant_step_level(api, oracle_built, case):
    """One sweep + every trial of the ladder for the best-merit / heun / rk3 variants: K, k, V_x, V_xx 1e-8 and
    identical trial records (the integrator only enters the rollout; the sweep linearises with Euler, Appendix A.1)."""
    p = VARIANTS[case](api)
    B = 8
    x0, U0, X0 = _inputs(api, p, B, 20260928)
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0, U0, X0); hs.initialize()
    ok = hs.backward()
    K, k = hs.gains(); Vx, Vxx = hs.value()
    alphas = api.Oracle(p).alphas()
    trials = hs.forward(alphas)
    hs.close()
    for b in range(B):
        o = api.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b]); o.initialize()
        assert o.backward(retry=True) == ok[b]
        Ko, ko = o.gains(); Vxo, Vxxo = o.value()
        assert max(rel_err(K[b], Ko), rel_err(k[b], ko), rel_err(Vx[b], Vxo), rel_err(Vxx[b], Vxxo)) < TOL
        for a, alpha in enumerate(alphas):
            t = o.forward(alpha); g = trials[b, a]
            assert g["success"] == t["success"], (case, b, alpha)
            assert abs(g["alpha_pr"] - t["alpha_pr"]) < 1e-9 and abs(g["alpha_du"] - t["alpha_du"]) < 1e-9
            if t["success"]:
                assert rel_err(g["cost"], t["cost"]) < TOL and rel_err(g["merit_function"], t["merit_function"]) < TOL


@pytest.mark.parametrize("case", list(VARIANTS))
def test_variant_full_solve(api, oracle_built, case):
    """Solve-level parity of the variants with the rule of test_full_solve_parity (no waiver): identical status and
    iteration count for EVERY trajectory, and every trajectory strict (work counts, objective 1e-7, X / U 1e-6)."""
    p = VARIANTS[case](api)
    B = 16
    res, X, U, K, ores, oX, oU, oK = _solve_both(api, p, B, 20260929)
    same_counts, same_work, strict, conv = _agreement(api, res, ores, X, oX, U, oU, K, oK)
    _report("variant_" + case, {"B": B, "same_counts": int(same_counts.sum()), "strict": int(strict.sum()), "converged": int(conv.sum()),
                                "mean_iterations": float(np.mean(res["iterations"]))})
    assert same_counts.all(), (case, list(zip(res["iterations"], ores["iterations"], res["status"], ores["status"])))
    assert strict[conv].all(), (case, strict, conv)
    assert strict[0] and strict.sum() == B, (case, strict)   # round 4: same arithmetic on both sides, every trajectory strict


@pytest.mark.parametrize("case,parallel,mask", [("cartpole_ipddp_box", True, 1), ("cartpole_ipddp_box", True, 5), ("unicycle_ipddp_box_ball", True, 3),
                                                ("cartpole_ipddp_box", False, 1), ("pendulum_ipddp_box", True, 1)])
def test_discarded_candidate_moves_to_the_next_trial(api, oracle_built, case, parallel, mask, monkeypatch):
    """The reference's parallel rule DISCARDS a forward pass that threw and keeps the best of the others
    (cddp_solver_base.cpp:264-314; pinned by its ParallelForwardPassKeepsSuccessfulAlphaWhenAnotherThrows with a mock solver whose
    alpha = 1 throws, tests/cddp_core/test_cddp_core.cpp:414-435); in IPDDP a trial whose costate is not finite fails the same way
    (ipddp_solver.cpp:1613-1616).  On the device that is k_update's candidate walk: the least-merit trial carries flag 2 ("passed every
    other test, costate not finite"), the remaining successful trials are walked in merit order and their costate is evaluated by
    costate_trial_serial.  No finite problem reaches that branch on its own (V_xx dx overflows only after the cost does), so both sides
    get the same mock: alpha indices in `mask` are evaluated and discarded (CDDP_HIP_TEST_FAIL_COSTATE / oracle set_failing_alphas).
    Every trajectory: status, iterations, sweeps, rollouts identical and the iterate strict; and the mock did change the solve."""
    p = (_parallel(api, case) if parallel else make(api, case))
    B = 16
    x0, U0, X0 = _inputs(api, p, B, 20260929)

    def hip():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0, X0); hs.solve()
        out = (hs.results().copy(),) + hs.trajectory() + hs.gains()[:1]; hs.close()
        return out

    plain = hip()[0]
    monkeypatch.setenv("CDDP_HIP_TEST_FAIL_COSTATE", str(mask))
    api.set_failing_alphas(mask)
    try:
        res, X, U, K = hip()
        ores, oX, oU, oK, _ = api.oracle_solve_batch(p, x0, U0, X0, n_threads=8)
    finally:
        api.set_failing_alphas(0)
    same_counts, same_work, strict, conv = _agreement(api, res, ores, X, oX, U, oU, K, oK)
    _report("discarded_%s_%s_%d" % (case, "parallel" if parallel else "first", mask),
            {"B": B, "same_counts": int(same_counts.sum()), "same_work": int(same_work.sum()), "strict": int(strict.sum()),
             "changed_vs_plain": int(np.sum((res["n_forward"] != plain["n_forward"]) | (res["final_objective"] != plain["final_objective"])))})
    assert same_counts.all() and same_work.all(), (case, list(zip(res["iterations"], ores["iterations"], res["n_forward"], ores["n_forward"])))
    assert strict.all(), (case, strict)
    assert np.any((res["n_forward"] != plain["n_forward"]) | (res["final_objective"] != plain["final_objective"]) | (res["iterations"] != plain["iterations"]))


def test_best_merit_differs_from_first_success(api):
    """The two rules are different algorithms: on the cart-pole they must not be silently
    the same code path -- at least one trajectory accepts a different trial sequence."""
    p1 = make(api, "cartpole_ipddp_box"); p2 = _parallel(api, "cartpole_ipddp_box")
    B = 16
    x0, U0, X0 = _inputs(api, p1, B, 20260929)
    out = []
    for p in (p1, p2):
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0, X0); hs.solve(); out.append(hs.results().copy()); hs.close()
    assert np.any(out[0]["n_forward"] != out[1]["n_forward"]) or np.any(out[0]["final_objective"] != out[1]["final_objective"])


# ------------------------------------------------------------------------------------------------------------------
# (d) late-iterate gains: the oracle's iterate k is installed in the handle, ONE backward pass on both sides
# ------------------------------------------------------------------------------------------------------------------
LATE = [("cartpole_ipddp_box", 10), ("cartpole_ipddp_box", 40), ("cartpole_ipddp_box", 79),
        ("unicycle_ipddp_box_ball", 10), ("unicycle_ipddp_box_ball", 40), ("unicycle_ipddp_box_ball", 90),
        ("pendulum_ipddp_box", 8), ("pendulum_ipddp_box", 20),
        ("cartpole_ipddp_box_state", 30), ("quad12_ipddp_box", 25), ("manip7_ipddp_box", 25)]


@pytest.mark.parametrize("case,kit", LATE)
def test_late_iterate_gains(api, oracle_built, case, kit):
    """K, k, V_x, V_xx, dV of the sweep at iterate `kit` (small mu, Y S^-1 near the 1e6 clip, regularisation off its
    initial value) within 1e-8 of the oracle's sweep from the SAME iterate: X, U, S, Y, mu, reg of the oracle after
    `kit` iterations are copied into the handle (set_initial / set_duals / set_barrier_state)."""
    p = make(api, case)
    p.options.max_iterations = kit
    B = 6 if p.nx <= 4 else 3
    x0, U0, X0 = _inputs(api, p, B, 20260930)
    Xk = np.zeros((B, p.N + 1, p.nx)); Uk = np.zeros((B, p.N, p.nu))
    m = p.dual_dim()
    Sk = np.zeros((B, p.N, m)); Yk = np.zeros((B, p.N, m)); mu = np.zeros(B); reg = np.zeros(B)
    orc = []
    for b in range(B):
        o = api.Oracle(p)
        o.set_initial(x0[b], None if U0 is None else U0[b], None if X0 is None else X0[b])
        r = o.solve()
        Xk[b], Uk[b] = o.trajectory()
        Sk[b], Yk[b], _ = o.duals()
        mu[b] = r["barrier_mu"]; reg[b] = r["regularization"]
        orc.append((o, r))
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(np.ascontiguousarray(Xk[:, 0, :]), Uk, Xk)
    hs.initialize()                       # cold: re-rolls X from U (ipddp_solver.cpp:868-874), evaluates g
    hs.set_duals(Sk, Yk)
    hs.set_barrier_state(mu, reg)
    ok = hs.backward()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg_after = hs.backward_scalars()
    Xd, _ = hs.trajectory()
    hs.close()
    worst = {"K": 0.0, "k": 0.0, "Vx": 0.0, "Vxx": 0.0, "dV": 0.0}
    ys_max = 0.0
    for b in range(B):
        o, r = orc[b]
        if r["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE, api.STATUS_REG_LIMIT):
            continue                      # finished before iterate `kit`: its last iterate is still a valid state
        assert rel_err(Xd[b], Xk[b]) < 1e-9, "re-rolled trajectory differs from the oracle's iterate"
        ook = o.backward(retry=True)
        assert ok[b] == ook, (case, kit, b)
        if not ook:
            continue
        Ko, ko = o.gains(); Vxo, Vxxo = o.value(); dVo, rego = o.backward_scalars()
        e = {"K": rel_err(K[b], Ko), "k": rel_err(k[b], ko), "Vx": rel_err(Vx[b], Vxo), "Vxx": rel_err(Vxx[b], Vxxo), "dV": rel_err(dV[b], dVo)}
        for key in worst:
            worst[key] = max(worst[key], e[key])
        assert reg_after[b] == rego
        with np.errstate(divide="ignore", invalid="ignore"):
            ys_max = max(ys_max, float(np.max(Yk[b] / np.maximum(Sk[b], 1e-300))))
    _report("late_%s_%d" % (case, kit), {"worst_rel_err": worst, "mu": [float(v) for v in mu], "reg": [float(v) for v in reg], "max_y_over_s": ys_max})
    assert max(worst.values()) < TOL, (case, kit, worst)


# ------------------------------------------------------------------------------------------------------------------
# (e) knife-edge plants: measured flip rate over a batch of 32
# ------------------------------------------------------------------------------------------------------------------
# Yardstick: tests/golden/trig_noise_flip_rates.json -- how many of these 32 trajectories keep (status, iterations) when
# the ORACLE's own sin / cos results are moved by <= 1 ulp (tests/golden/make_trig_noise.py), i.e. under the difference
# between glibc and the device libm.  A capped trial lands exactly on (1 - tau) s, so `s_new < (1 - tau) s` is decided by
# the last bit; after a flip two solves follow different, equally valid iterates.  The HIP path may not flip more often
# than that noise does (margin: 3 of 32 trajectories, about one binomial standard deviation at the observed rates; the strict,
# non-statistical comparison of the same cases is tests/test_shared_trig_parity.py).
KNIFE_MARGIN = 3
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trig_noise_flip_rates.json")) as _f:
    TRIG_NOISE = json.load(_f)


@pytest.mark.parametrize("case", sorted(KNIFE_EDGE_CASES))
def test_knife_edge_flip_rate(api, oracle_built, case):
    p = TERM_CASES[case](api) if case in TERM_CASES else make(api, case)
    B = 32
    res, X, U, K, ores, oX, oU, oK = _solve_both(api, p, B, 20260929)
    same_counts, same_work, strict, conv = _agreement(api, res, ores, X, oX, U, oU, K, oK)
    both_conv = conv & ((res["status"] == api.STATUS_OPTIMAL) | (res["status"] == api.STATUS_ACCEPTABLE))
    obj_err = [rel_err(res["final_objective"][b], ores["final_objective"][b]) for b in range(B) if both_conv[b]]
    _report("knife_" + case, {"B": B, "same_counts": int(same_counts.sum()), "same_work": int(same_work.sum()), "strict": int(strict.sum()),
                              "converged_oracle": int(conv.sum()), "flip_rate": float(1.0 - same_counts.mean()),
                              "oracle_trig_noise_same_counts": TRIG_NOISE[case]["same_counts"],
                              "max_objective_rel_err_converged": float(max(obj_err)) if obj_err else 0.0})
    # Round 4: the shipped library runs the oracle's arithmetic (trig_mode 1 in every gpu test), so the yardstick of rounds 2-3 (the
    # oracle's own agreement under 1-ulp trig noise, minus a margin) is replaced by the strict statement: no trajectory flips.
    assert same_counts.sum() == B and same_work.sum() == B, (case, int(same_counts.sum()), int(same_work.sum()), TRIG_NOISE[case])
    for b in range(B):   # where both converge they converge to the same optimum
        if both_conv[b]:
            assert rel_err(res["final_objective"][b], ores["final_objective"][b]) < 1e-4, (case, b)
