"""GPU parity tests: the HIP C-ABI path vs the CPU oracle on the same seeded inputs.

Tolerances (north_star): gains K, k and value V_x, V_xx within 1e-8 (relative to max(1,|ref|));
identical iteration counts and termination status on pendulum / cartpole.
"""
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-8


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    a = np.where(same_inf, 0.0, a); b = np.where(same_inf, 0.0, b)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def make(api, name):
    S = api
    table = {
        "pendulum_ipddp_unc": lambda: S.pendulum_problem(S.SOLVER_IPDDP, False),
        "pendulum_ipddp_box": lambda: S.pendulum_problem(S.SOLVER_IPDDP, True),
        "pendulum_clddp_unc": lambda: S.pendulum_problem(S.SOLVER_CLDDP, False),
        "pendulum_clddp_box": lambda: S.pendulum_problem(S.SOLVER_CLDDP, True),
        "cartpole_ipddp_unc": lambda: S.cartpole_problem(S.SOLVER_IPDDP, False),
        "cartpole_ipddp_box": lambda: S.cartpole_problem(S.SOLVER_IPDDP, True),
        "cartpole_clddp_unc": lambda: S.cartpole_problem(S.SOLVER_CLDDP, False),
        "cartpole_clddp_box": lambda: S.cartpole_problem(S.SOLVER_CLDDP, True),
        "unicycle_ipddp_box_ball": lambda: S.unicycle_problem(S.SOLVER_IPDDP, 100, True),
        "unicycle_ipddp_box": lambda: S.unicycle_problem(S.SOLVER_IPDDP, 100, False),
        "unicycle_clddp_box": lambda: _renamed_unicycle(S),
        "quadrotor_ipddp_box": lambda: S.quadrotor_problem(S.SOLVER_IPDDP, 30, True),
        "quadrotor_clddp_box": lambda: S.quadrotor_problem(S.SOLVER_CLDDP, 30, True),
        "quad12_ipddp_box": lambda: S.quadrotor12_problem(S.SOLVER_IPDDP, 30, True),
        "manipulator_clddp_box": lambda: S.manipulator_problem(S.SOLVER_CLDDP, 40, False, True),
        "manipulator_ipddp_box": lambda: S.manipulator_problem(S.SOLVER_IPDDP, 40, False, True),
        "manip7_ipddp_box": lambda: _manip7(S),
        "pendulum_ipddp_box_state": lambda: _with_state_box(S.pendulum_problem(S.SOLVER_IPDDP, True), [-4.0, -9.0], [4.0, 9.0]),
        "cartpole_ipddp_box_state": lambda: _with_state_box(S.cartpole_problem(S.SOLVER_IPDDP, True), [-1.5, -7.0, -8.0, -25.0], [1.5, 7.0, 8.0, 25.0]),
        "unicycle_ipddp_box_state": lambda: _with_state_box(S.unicycle_problem(S.SOLVER_IPDDP, 100, False), [-0.5, -0.5, -4.0], [2.6, 2.6, 4.0],
                                                            name="state_limits"),
        # f3 tail: ground-vehicle plants, cone / thrust-magnitude rows
        "bicycle_ipddp_box": lambda: S.bicycle_problem(S.SOLVER_IPDDP),
        "bicycle_ipddp_box_rk4": lambda: S.bicycle_problem(S.SOLVER_IPDDP, integrator=S.RK4),
        "bicycle_clddp_box": lambda: S.bicycle_problem(S.SOLVER_CLDDP),
        "hcw_ipddp_box": lambda: S.hcw_problem(S.SOLVER_IPDDP),
        "hcw_clddp_box": lambda: S.hcw_problem(S.SOLVER_CLDDP, integrator=S.EULER),
        "car_ipddp_box": lambda: S.car_problem(S.SOLVER_IPDDP),
        "car_clddp_box": lambda: S.car_problem(S.SOLVER_CLDDP),
        "unicycle_ipddp_box_soc": lambda: S.unicycle_cone_problem(S.SOLVER_IPDDP),
        "unicycle_ipddp_thrust": lambda: S.unicycle_thrust_problem(S.SOLVER_IPDDP, two_sided=True),
        "unicycle_ipddp_maxthrust": lambda: S.unicycle_thrust_problem(S.SOLVER_IPDDP, two_sided=False),
    }
    if name in OPTION_CASES:
        return OPTION_CASES[name](S)
    if name in TERM_CASES:
        return TERM_CASES[name](S)
    return table[name]()


def _with_state_box(p, lo, hi, name="StateConstraint"):
    """StateConstraint next to the control box (sorted after it by name): rows that read x (G_x blocks)."""
    p.add_state_box(name, lo, hi)
    return p


def _renamed_unicycle(S):
    """CLDDP honours the box only when it is literally named 'ControlConstraint' (clddp_solver.cpp:85-86)."""
    p = S.unicycle_problem(S.SOLVER_CLDDP, 100, False)
    p._cons[0].name = b"ControlConstraint"; p._rebuild()
    return p


def _manip7(S):
    p = S.manipulator7_problem(S.SOLVER_IPDDP, 30, terminal_equality=False, n_alphas=11)
    p.options.enable_parallel = 0
    return p


def _lti(S, horizon, x0, goal, R, Qf, o):
    p = S.Problem(S.SOLVER_IPDDP, S.MODEL_LTI, S.EULER, 1, 1, horizon, 1.0, np.zeros((1, 1)), R * np.eye(1), Qf * np.eye(1), [goal],
                  lti_A=np.eye(1), lti_B=np.eye(1), options=o)
    p.x0 = np.array([x0])
    return p


def _term_opts(S, max_it=100):
    o = S.default_options(); o.max_iterations = max_it; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-6
    o.reg_initial_value = 1e-6; o.barrier_mu_initial = 1e-1
    return o


def _term_ineq_only(S):     # tests/cddp_core/test_ipddp_solver.cpp:1147-1207
    p = _lti(S, 8, 0.0, 1.0, 1e-2, 100.0, _term_opts(S, 60))
    p.add_terminal_inequality("TerminalUpperBound", np.eye(1), np.zeros(1))
    return p


def _term_eq_only(S):       # :1580-1637
    o = _term_opts(S); o.barrier_mu_initial = 1.0
    p = _lti(S, 8, 1.0, 0.0, 1e-2, 1.0, o)
    p.add_terminal_equality("TerminalTarget", [0.0])
    return p


def _path_term_eq(S):       # :1382-1438
    p = _lti(S, 8, 1.0, 0.0, 1e-2, 0.0, _term_opts(S))
    p.add_linear("LoosePathUpperBound", np.eye(1), [10.0])
    p.add_terminal_equality("TerminalTarget", [0.0])
    return p


def _path_term_ineq(S):     # makeScalarIntegratorProblem(path, terminal inequality) :156-207
    o = _term_opts(S, 20)
    p = _lti(S, 4, 1.0, 0.0, 1e-2, 1.0, o)
    p.add_linear("PathUpperBound", np.eye(1), [0.25])
    p.add_terminal_inequality("TerminalUpperBound", np.eye(1), [0.25])
    return p


def _pendulum_term(S):
    p = S.pendulum_problem(S.SOLVER_IPDDP, True, horizon=60)
    p.add_terminal_equality("TerminalTarget", [0.0, 0.0])
    return p


def _manip_term(S):
    p = S.manipulator_problem(S.SOLVER_IPDDP, 30, True, True)
    return p


def _manip7_term(S):
    p = S.manipulator7_problem(S.SOLVER_IPDDP, 20, terminal_equality=True, n_alphas=16)
    return p


def _manip7_term_short(S):
    # four steps of 10 ms cannot carry the arm to the goal: A_s (sensitivity of x_T to the terminal multipliers) is badly
    # conditioned -- the reduced system's pivoting, regularisation scales and step cap (ipddp_solver.cpp:550-617) decide the step
    p = S.manipulator7_problem(S.SOLVER_IPDDP, 4, terminal_equality=True, n_alphas=16)
    p.options.max_iterations = 12
    return p


def _pendulum_term_one_step(S):
    # one control for two terminal rows: A_s^T A_s is singular, only the regularisation shift makes the factorisation go through
    p = S.pendulum_problem(S.SOLVER_IPDDP, True, horizon=1)
    p.add_terminal_equality("TerminalTarget", [0.0, 0.0])
    p.options.max_iterations = 12
    return p


TERM_CASES = {"term_ineq_only": _term_ineq_only, "term_eq_only": _term_eq_only, "path_term_eq": _path_term_eq,
              "path_term_ineq": _path_term_ineq, "pendulum_term_eq": _pendulum_term, "manipulator_term_eq": _manip_term,
              "manip7_term_eq_parallel_ls": _manip7_term, "manip7_term_eq_short": _manip7_term_short,
              "pendulum_term_eq_one_step": _pendulum_term_one_step}


# ---- option branches of the ABI that have device code of their own: every one gets the step-level
# and solve-level parity tests below, a twin fixture (tests/golden/twin_<name>.json) and, where the reference pins the
# behaviour, its pin (tests/test_option_branches.py)
BARRIER_MONOTONIC, BARRIER_IPOPT = 1, 2


def _opt_variant(S, base, **kw):
    p = make(S, base)
    for k, v in kw.items():
        setattr(p.options, k, v)
    return p


OPTION_CASES = {
    # updateBarrierParameters' non-ADAPTIVE branch (ipddp_solver.cpp:2601-2614): MONOTONIC and IPOPT share it
    "cartpole_ipddp_box_monotonic": lambda S: _opt_variant(S, "cartpole_ipddp_box", barrier_strategy=BARRIER_MONOTONIC),
    "pendulum_ipddp_box_ipopt": lambda S: _opt_variant(S, "pendulum_ipddp_box", barrier_strategy=BARRIER_IPOPT),
    "unicycle_ipddp_box_ball_ipopt": lambda S: _opt_variant(S, "unicycle_ipddp_box_ball", barrier_strategy=BARRIER_IPOPT),
    # computeTheta with theta_norm = "l2" (ipddp_solver.cpp:2778-2848)
    "pendulum_ipddp_box_l2": lambda S: _opt_variant(S, "pendulum_ipddp_box", ipddp_theta_norm_l2=1),
    "unicycle_ipddp_box_ball_l2": lambda S: _opt_variant(S, "unicycle_ipddp_box_ball", ipddp_theta_norm_l2=1),
    "path_term_ineq_l2": lambda S: _opt_variant(S, "path_term_ineq", ipddp_theta_norm_l2=1),
    # computeScaledDualInfeasibility with check_state_stationarity (ipddp_solver.cpp:931, 2725-2776) on state-dependent rows
    "path_term_ineq_stationarity": lambda S: _opt_variant(S, "path_term_ineq", ipddp_check_state_stationarity=1),
    "pendulum_ipddp_box_state_stationarity": lambda S: _opt_variant(S, "pendulum_ipddp_box_state", ipddp_check_state_stationarity=1),
    "cartpole_ipddp_box_state_stationarity": lambda S: _opt_variant(S, "cartpole_ipddp_box_state", ipddp_check_state_stationarity=1),
}

# plants whose dynamics call sin/cos (device libm vs glibc differ in the last bit) AND whose caps bind
KNIFE_EDGE_CASES = {"manipulator_term_eq", "manip7_term_eq_parallel_ls", "manip7_ipddp_box", "manipulator_ipddp_box",
                    "quadrotor_ipddp_box", "quad12_ipddp_box"}

BIG_CASES = ["unicycle_clddp_box", "quadrotor_ipddp_box", "quadrotor_clddp_box", "quad12_ipddp_box",
             "manipulator_clddp_box", "manipulator_ipddp_box", "manip7_ipddp_box"]

# f3 tail: car / bicycle plants, cone and thrust-magnitude rows.  Their solves run on the parity build (shared sin / cos / asin / tan):
# asin and tan are two more libm routines whose device and host versions agree to an ulp, not to the bit.
F3_CASES = ["bicycle_ipddp_box", "bicycle_ipddp_box_rk4", "bicycle_clddp_box", "hcw_ipddp_box", "hcw_clddp_box", "car_ipddp_box", "car_clddp_box",
            "unicycle_ipddp_box_soc", "unicycle_ipddp_thrust", "unicycle_ipddp_maxthrust"]

CASES = ["pendulum_ipddp_unc", "pendulum_ipddp_box", "pendulum_clddp_unc", "pendulum_clddp_box",
         "cartpole_ipddp_unc", "cartpole_ipddp_box", "cartpole_clddp_unc", "cartpole_clddp_box",
         "unicycle_ipddp_box", "unicycle_ipddp_box_ball",
         "pendulum_ipddp_box_state", "cartpole_ipddp_box_state", "unicycle_ipddp_box_state"]


def spread_for(p):
    s = 0.1 * np.ones(p.nx)
    if p.nx >= 6:
        s[:] = 0.02
    if p.nx == 4:
        s[1] = 0.3
    if p.nx == 3:
        s[:] = 0.05
    return s


@pytest.mark.parametrize("case", CASES + BIG_CASES + list(TERM_CASES) + list(OPTION_CASES) + F3_CASES)
def test_step_level_parity(api, oracle_built, case):
    """initialize -> backward -> forward(alphas): K, k, V_x, V_xx, dV and every trial record."""
    p = make(api, case)
    B = 8 if p.nx <= 4 else 3
    x0 = api.batch_x0(p, B, 20260928, spread_for(p))
    U0 = api.batch_U0(p, B)
    X0 = np.tile(p.X0_single, (B, 1, 1)) if hasattr(p, "X0_single") else None
    if X0 is not None:
        X0[:, 0, :] = x0
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0, U0, X0)
    hs.initialize()
    ok = hs.backward()
    K, k = hs.gains()
    Vx, Vxx = hs.value()
    dV, reg = hs.backward_scalars()
    alphas = api.Oracle(p).alphas()
    trials = hs.forward(alphas)
    for b in range(B):
        o = api.Oracle(p)
        o.set_initial(x0[b], None if U0 is None else U0[b], None if X0 is None else X0[b])
        o.initialize()
        ook = o.backward(retry=True)
        assert ok[b] == ook
        Ko, ko = o.gains(); Vxo, Vxxo = o.value(); dVo, rego = o.backward_scalars()
        assert rel_err(K[b], Ko) < TOL, (case, b, rel_err(K[b], Ko))
        assert rel_err(k[b], ko) < TOL
        assert rel_err(Vx[b], Vxo) < TOL
        assert rel_err(Vxx[b], Vxxo) < TOL
        assert rel_err(dV[b], dVo) < TOL
        assert reg[b] == rego
        for a, alpha in enumerate(alphas):
            t = o.forward(alpha)
            g = trials[b, a]
            assert abs(g["alpha_pr"] - t["alpha_pr"]) < 1e-9 and abs(g["alpha_du"] - t["alpha_du"]) < 1e-9
            # Round 4: the library evaluates the plants' sin / cos and the core's log / pow with the routines the oracle runs here
            # (trig_mode 1, tests/conftest.py), so even a CAPPED trial -- one that lands exactly on the fraction-to-boundary bound
            # (1 - tau) s, decided by the last bit -- takes the oracle's decision.  (Rounds 1-3 allowed such flips on the knife-edge
            # plants: device libm against glibc.)
            assert g["success"] == t["success"], (case, b, alpha, g, t)
            if t["success"]:
                assert rel_err(g["cost"], t["cost"]) < TOL
                assert rel_err(g["merit_function"], t["merit_function"]) < TOL
                assert rel_err(g["theta"], t["theta"]) < TOL
    hs.close()


@pytest.mark.parametrize("case", CASES + list(TERM_CASES) + list(OPTION_CASES) + F3_CASES)
def test_full_solve_parity(api, oracle_built, case):
    """cddp_hip_solve vs oracle solve: identical iteration counts / status; trajectories, gains within 1e-8...
    (full trajectories pass through up to 80 nonlinear iterations, so they are compared at 1e-6)."""
    p = make(api, case)
    p.options.return_iteration_info = 1
    B = 16 if p.nx <= 4 else 4
    x0 = api.batch_x0(p, B, 20260929, spread_for(p))
    U0 = api.batch_U0(p, B)
    X0 = np.tile(p.X0_single, (B, 1, 1)) if hasattr(p, "X0_single") else None
    if X0 is not None:
        X0[:, 0, :] = x0
    # Round 4: ONE library, built with the shared straight-line arithmetic, against the oracle in the same arithmetic
    # (tests/conftest.py sets trig_mode 1 for every gpu test): most of these cases never converge inside the iteration cap (the
    # cart-pole example itself does not), and a non-converging solve is a chaotic map of the last bit of every sine -- with the
    # same routines on both sides the comparison is strict for EVERY case, the knife-edge plants included.
    shared = True
    octx = contextlib.nullcontext
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0, U0, X0)
    st = hs.solve()
    res = hs.results()
    X, U = hs.trajectory()
    K, k = hs.gains()
    hist = hs.history(B)
    with octx():
        ores, oX, oU, oK, _ = api.oracle_solve_batch(p, x0, U0, X0, n_threads=8)
    mism = [(b, int(res["iterations"][b]), int(ores["iterations"][b]), int(res["status"][b]), int(ores["status"][b]))
            for b in range(B) if res["iterations"][b] != ores["iterations"][b] or res["status"][b] != ores["status"][b]]
    assert not mism, (case, mism)
    # Every trajectory must agree in every counter and to 1e-6 in the trajectories (`shared`: the library and the checker run the same
    # sin / cos / log / pow and neither contracts to FMA, so even the solves that end at MaxIterations / RegularizationLimit -- chaotic
    # maps of their rounding -- take the same line-search decisions).  Only a checker in glibc arithmetic (`shared` false, a hand-made
    # comparison build) is allowed 10 % of non-converged trajectories that part ways after dozens of iterations.
    conv = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    strict = np.ones(B, dtype=bool)
    for b in range(B):
        same = (res["n_backward"][b] == ores["n_backward"][b] and res["n_forward"][b] == ores["n_forward"][b]
                and rel_err(res["final_objective"][b], ores["final_objective"][b]) < 1e-7
                and rel_err(X[b], oX[b]) < 1e-6 and rel_err(U[b], oU[b]) < 1e-6 and rel_err(K[b], oK[b]) < 1e-5)
        strict[b] = same
        if conv[b]:
            assert same, (case, b, res[b], ores[b], rel_err(X[b], oX[b]), rel_err(K[b], oK[b]))
        # non-converged: status + iteration count already compared above
    assert strict[0], "trajectory 0 (the reference example) must match strictly"
    assert strict.sum() >= (B if shared else int(np.ceil(0.9 * B))), (case, strict)
    # per-iteration trace of trajectory 0 (the unperturbed reference example)
    with octx():
        o = api.Oracle(p); o.set_initial(x0[0], None if U0 is None else U0[0], None if X0 is None else X0[0]); o.solve()
    oh = o.history()
    assert hist[0].shape == oh.shape, (hist[0].shape, oh.shape)
    assert rel_err(hist[0], oh) < 1e-6
    assert st.n_converged == int(np.sum((ores["status"] == 1) | (ores["status"] == 2)))
    if p.c.solver == api.SOLVER_IPDDP:   # the costate rows K4b writes for the accepted trial (cddp_hip_get_costates, round 5): N + 1 rows
        L = hs.costates(); oL = o.costates()
        assert L.shape == (B,) + oL.shape and oL.shape == (p.N + 1, p.nx), (L.shape, oL.shape)
        assert rel_err(L[0], oL) < 1e-6, (case, rel_err(L[0], oL))
    hs.close()


@pytest.mark.parametrize("case", ["cartpole_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_ipddp_box"])
def test_multi_alpha_rollout_groups_agree_bitwise(api, case, monkeypatch):
    """kernels_pcm.hpp (opt-in, CDDP_HIP_K4_NA = 2 | 3): NA producer waves + one consumer wave per (tile, group of NA step sizes)
    must leave every iterate, gain and counter exactly as the two-wave rollout does -- incl. ladders whose last group is short
    (11 step sizes: groups of 3 + 3 + 3 + 2 and 2 x 5 + 1) and the two-stage ladder."""
    p = make(api, case)
    B = 96
    x0 = api.batch_x0(p, B, 20261103, spread_for(p))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); S, Y, G = hs.duals(); hs.close()
        return [r[f].copy() for f in r.dtype.names] + [X, U, K, k, S, Y, G, np.array([st.sweeps, st.rollouts, st.rollout_steps])]

    monkeypatch.delenv("CDDP_HIP_K4_NA", raising=False)
    ref = run()
    for na, stages in (("2", None), ("3", None), ("3", "2")):
        monkeypatch.setenv("CDDP_HIP_K4_NA", na)
        if stages:
            monkeypatch.setenv("CDDP_HIP_LS_STAGES", stages)
        got = run()
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), (case, na, stages)


@pytest.mark.parametrize("case", ["unicycle_ipddp_box_ball", "unicycle_ipddp_box"])
def test_two_consumer_rollout_agrees_bitwise(api, case, monkeypatch):
    """Round 5 (opt-in, CDDP_HIP_K4_CONSUMERS=2; measured no gain inside the solve): TWO consumer waves per (tile, step size) for the unicycle
    layouts (k_forward_ipddp_pc<.., NC = 2>: consumer c takes the steps
    t = c mod 2, the producer sums the running cost, the first object's |g + s| terms are parked and summed in t order afterwards, the
    waves' maxima / first failing steps are merged).  Against the default one-consumer kernel: every result word,
    iterate, gain, dual row and the work counters (incl. the credited rollout steps, i.e. each trial's first failing step) are the same
    bits -- under both ladder shapes, for a batch with a ragged last tile, and for the trial records of a step-level forward pass."""
    p = make(api, case)
    B = 150
    x0 = api.batch_x0(p, B, 20270305, spread_for(p))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); S, Y, G = hs.duals(); hs.close()
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize(); hs.backward()
        tr = hs.forward(api.Oracle(p).alphas()); hs.close()
        return [r[f].copy() for f in r.dtype.names] + [X, U, K, k, S, Y, G, np.array([st.sweeps, st.rollouts, st.rollout_steps])] + [tr[f].copy() for f in tr.dtype.names]

    monkeypatch.delenv("CDDP_HIP_K4_CONSUMERS", raising=False)
    monkeypatch.delenv("CDDP_HIP_LS_STAGES", raising=False)
    ref = run()
    for stages in (None, "1", "2"):
        monkeypatch.setenv("CDDP_HIP_K4_CONSUMERS", "2")
        if stages: monkeypatch.setenv("CDDP_HIP_LS_STAGES", stages)
        got = run()
        for i, (a_, b_) in enumerate(zip(ref, got)):
            assert np.array_equal(a_, b_, equal_nan=(a_.dtype.kind == "f")), (case, stages, i)


@pytest.mark.parametrize("case", ["cartpole_ipddp_box", "unicycle_ipddp_box_ball", "cartpole_clddp_box"])
def test_two_stage_ladder_selects_the_same_trials(api, case, monkeypatch):
    """The speculative single-launch ladder and the two-stage ladder (alpha_0, then the rest for the trajectories
    that need them; used when batch x n_alpha overfills the chip) must accept the very same trials."""
    p = make(api, case)
    B = 96
    x0 = api.batch_x0(p, B, 20260931, spread_for(p))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); hs.close()
        return r, X, U, K, k

    monkeypatch.delenv("CDDP_HIP_LS_STAGES", raising=False)
    r1, X1, U1, K1, k1 = run()
    monkeypatch.setenv("CDDP_HIP_LS_STAGES", "2")
    r2, X2, U2, K2, k2 = run()
    assert np.array_equal(r1["iterations"], r2["iterations"]) and np.array_equal(r1["status"], r2["status"])
    assert np.array_equal(r1["final_objective"], r2["final_objective"])
    assert np.array_equal(X1, X2) and np.array_equal(U1, U2) and np.array_equal(K1, K2) and np.array_equal(k1, k2)
    monkeypatch.setenv("CDDP_HIP_LS_FIRST", "3")      # stage 1 = the first three alphas, stage 2 = the rest
    r3, X3, U3, K3, k3 = run()
    assert np.array_equal(r1["iterations"], r3["iterations"]) and np.array_equal(r1["status"], r3["status"])
    assert np.array_equal(X1, X3) and np.array_equal(U1, U3) and np.array_equal(K1, K3) and np.array_equal(k1, k3)


@pytest.mark.parametrize("case", ["term_eq_only", "path_term_eq", "pendulum_term_eq", "manipulator_term_eq",
                                  "manip7_term_eq_parallel_ls", "cartpole_ipddp_box", "quadrotor_ipddp_box",
                                  "cartpole_clddp_box", "pendulum_clddp_box"])   # CLDDP nu = 1: straight-line BoxQP (coop) vs its loop (lane)
def test_cooperative_and_lane_sweeps_agree_bitwise(api, case, monkeypatch):
    """The lane-cooperative sweeps (kernels_coop.hpp, kernels_te.hpp: column per lane, gradient variant per lane,
    two-role rollout) restate the one-lane-per-trajectory kernels sum for sum; CDDP_HIP_SWEEP=lane selects the
    latter.  Iterates, gains and counters of both must be the SAME bits (terminal-equality cases: the whole
    reduced-LQR branch incl. the multiplier step)."""
    p = TERM_CASES[case](api) if case in TERM_CASES else make(api, case)
    B = 70
    x0 = api.batch_x0(p, B, 20261005, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); hs.close()
        return r, X, U, K, k

    monkeypatch.delenv("CDDP_HIP_SWEEP", raising=False)
    r1, X1, U1, K1, k1 = run()
    monkeypatch.setenv("CDDP_HIP_SWEEP", "lane")
    r2, X2, U2, K2, k2 = run()
    assert np.array_equal(r1["iterations"], r2["iterations"]) and np.array_equal(r1["status"], r2["status"])
    assert np.array_equal(r1["final_objective"], r2["final_objective"])
    assert np.array_equal(X1, X2) and np.array_equal(U1, U2) and np.array_equal(K1, K2) and np.array_equal(k1, k2)


ELEM_CASES = ["pendulum_ipddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_ipddp_box_state",
              "cartpole_ipddp_box_state", "unicycle_ipddp_box_state"]


@pytest.mark.parametrize("case", ELEM_CASES)
def test_element_sweep_agrees_bitwise(api, case, monkeypatch):
    """nx <= 4, nu <= 2 (round 3): the column-ownership sweep (the default; exchanges by quad broadcast), the element-ownership sweep
    (kernels_elem.hpp: lane (i, j) owns V_xx[i][j]; CDDP_HIP_SWEEP=elem) and the one-lane-per-trajectory sweep
    (CDDP_HIP_SWEEP=lane) give the same bits -- whole solves, every layout with nx in {2, 3, 4}, nu in {1, 2}, with and without
    state-dependent rows; a batch that is not a multiple of the 4 trajectories of a wavefront; gains / value function of the
    first sweep as well."""
    p = make(api, case)
    B = 70 + 3
    x0 = api.batch_x0(p, B, 20261006, spread_for(p))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize(); ok = hs.backward()
        K0, k0 = hs.gains(); Vx0, Vxx0 = hs.value(); dV0, reg0 = hs.backward_scalars()
        hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); Vx, Vxx = hs.value(); hs.close()
        return (ok, K0, k0, Vx0, Vxx0, dV0, reg0, r["iterations"], r["status"], r["final_objective"], r["n_backward"], r["n_forward"], X, U, K, k, Vx, Vxx)

    monkeypatch.delenv("CDDP_HIP_SWEEP", raising=False)
    ref = run()
    for mode in ("elem", "lane"):
        monkeypatch.setenv("CDDP_HIP_SWEEP", mode)
        got = run()
        for a, g in zip(ref, got):
            assert np.array_equal(a, g), (case, mode)


@pytest.mark.parametrize("case", ["quad12_ipddp_box", "quadrotor_ipddp_box", "manip7_ipddp_box"])
def test_row_split_sweep_agrees_bitwise(api, case, monkeypatch):
    """nx > 8: the LDS-operand sweep with two lanes per column (32 lanes per trajectory, picked for small batches),
    with one lane per column, and the one-lane-per-trajectory kernel -- the same bits, whole solves."""
    p = make(api, case)
    B = 37
    x0 = api.batch_x0(p, B, 20261112, spread_for(p))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); Vx, Vxx = hs.value(); hs.close()
        return r, X, U, K, k, Vx, Vxx

    monkeypatch.delenv("CDDP_HIP_SWEEP", raising=False)
    monkeypatch.delenv("CDDP_HIP_COOP_H", raising=False)
    out = {}
    monkeypatch.setenv("CDDP_HIP_COOP_W", "1")          # one wavefront per group, one lane per column
    out["1"] = run()
    monkeypatch.setenv("CDDP_HIP_COOP_H", "2")          # one wavefront per group, two lanes per column
    out["2"] = run()
    monkeypatch.delenv("CDDP_HIP_COOP_H", raising=False)
    monkeypatch.delenv("CDDP_HIP_COOP_W", raising=False)
    out["two_wave"] = run()                             # the default (round 5): A side / gain side on two wavefronts
    monkeypatch.setenv("CDDP_HIP_SWEEP", "lane")
    out["lane"] = run()
    for other in ("2", "two_wave", "lane"):
        a, b = out["1"], out[other]
        assert np.array_equal(a[0]["iterations"], b[0]["iterations"]) and np.array_equal(a[0]["status"], b[0]["status"]), other
        assert np.array_equal(a[0]["final_objective"], b[0]["final_objective"]), other
        for key in a[0].dtype.names:
            assert np.array_equal(a[0][key], b[0][key], equal_nan=True), (other, key)
        for i in range(1, 7): assert np.array_equal(a[i], b[i]), (other, i)


@pytest.mark.parametrize("case,reg0", [("quad12_ipddp_box", 0.0), ("quad12_ipddp_box", 1e-9), ("manip7_ipddp_box", 0.0)])
def test_two_wave_sweep_agrees_bitwise(api, case, reg0, monkeypatch):
    """k_backward_ipddp_coop_big2 against the one-wave kernel on a batch whose trajectories do NOT move in step: a spread of
    initial states wide enough that some factorisations fail and restart their pass with more regularisation (the per-trajectory
    state machine of the two-wave kernel), a batch that leaves a partial group, and a short iteration budget so that trajectories
    stop in different states.  Result words, work counts, trajectories, gains and value function: the same bits."""
    p = make(api, case)
    p.options.max_iterations = 12
    if reg0 > 0.0: p.options.reg_initial_value = reg0
    B = 150
    x0 = api.batch_x0(p, B, 20270311, 6.0 * spread_for(p))
    U0 = api.batch_U0(p, B)

    def run():
        hs = api.HipBatchSolver(p, B)
        hs.set_initial(x0, U0); hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); Vx, Vxx = hs.value(); hs.close()
        return r, X, U, K, k, Vx, Vxx

    monkeypatch.delenv("CDDP_HIP_SWEEP", raising=False)
    monkeypatch.delenv("CDDP_HIP_COOP_H", raising=False)
    monkeypatch.setenv("CDDP_HIP_COOP_W", "1")
    a = run()
    monkeypatch.delenv("CDDP_HIP_COOP_W", raising=False)
    b = run()
    for key in a[0].dtype.names:
        assert np.array_equal(a[0][key], b[0][key], equal_nan=True), key
    for i in range(1, 7): assert np.array_equal(a[i], b[i], equal_nan=True), i
    print("backward passes per iteration (max over the batch):", np.max(a[0]["n_backward"] / np.maximum(1, a[0]["iterations"])))


@pytest.mark.parametrize("case", ["quad12_ipddp_box", "quadrotor_ipddp_box"])
def test_two_wave_sweep_restarts_agree_bitwise(api, case, monkeypatch):
    """The restart path of k_backward_ipddp_coop_big2: trajectories whose factorisation fails (a NaN control late in the horizon makes
    B_t and V_x non-finite) restart their pass with more regularisation, pass after pass up to the limit, while the other trajectories
    of the same group of four walk down the horizon -- every trajectory of a group at its own step.  Step level (initialize ->
    backward), against the one-wave kernel: verdicts, regularisation, pass counts, gains, value function, the same bits."""
    p = make(api, case)
    p.options.reg_initial_value = 1e2
    B = 23
    x0 = api.batch_x0(p, B, 20270312, spread_for(p))
    U0 = api.batch_U0(p, B)
    if U0 is None: U0 = np.zeros((B, p.N, p.nu))
    U0 = np.array(U0, copy=True)
    bad = [1, 6, 7, 13, 22]
    for b in bad: U0[b, p.N - 3, 0] = np.nan

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize()
        ok = hs.backward()
        K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars(); r = hs.results(); hs.close()
        return ok, K, k, Vx, Vxx, dV, reg, r["n_backward"]

    monkeypatch.delenv("CDDP_HIP_SWEEP", raising=False)
    monkeypatch.delenv("CDDP_HIP_COOP_H", raising=False)
    monkeypatch.setenv("CDDP_HIP_COOP_W", "1")
    a = run()
    monkeypatch.delenv("CDDP_HIP_COOP_W", raising=False)
    b = run()
    assert not a[0][bad].any() and a[0][[i for i in range(B) if i not in bad]].all(), a[0]
    assert (a[7][bad] > 1).all(), a[7]          # more than one pass: the restarts happened
    good = [i for i in range(B) if i not in bad]
    for i in range(8):
        assert np.array_equal(a[i][good], b[i][good]), i
    for i in (0, 6, 7):                          # verdict, regularisation, pass count of the failed ones (their stacks hold a failed pass)
        assert np.array_equal(a[i][bad], b[i][bad]), i


@pytest.mark.parametrize("sweep", ["coop", "lane"])
@pytest.mark.parametrize("case", ["cartpole_ipddp_box", "cartpole_ipddp_unc", "unicycle_ipddp_box_ball", "quadrotor_ipddp_box",
                                  "pendulum_term_eq", "manip7_term_eq_parallel_ls", "term_ineq_only"])
def test_value_hessian_is_stored_exactly_symmetric(api, case, sweep, monkeypatch):
    """k_costate fetches only the upper triangle of V_xx[t]: every IPDDP sweep must store V_xx bitwise symmetric
    (0.5 (M + M^T), sym(2 Q_f) at t = N), in the cooperative and in the one-lane kernels."""
    if sweep == "lane":
        monkeypatch.setenv("CDDP_HIP_SWEEP", "lane")
    else:
        monkeypatch.delenv("CDDP_HIP_SWEEP", raising=False)
    p = TERM_CASES[case](api) if case in TERM_CASES else make(api, case)
    B = 20
    x0 = api.batch_x0(p, B, 20261012, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, api.batch_U0(p, B)); hs.solve()
    _, Vxx = hs.value(); hs.close()
    assert np.all(np.isfinite(Vxx))
    assert np.array_equal(Vxx, np.swapaxes(Vxx, 2, 3))


@pytest.mark.parametrize("N", [1, 2, 3, 5])
@pytest.mark.parametrize("kind", ["scalar_path", "cartpole_box", "pendulum_clddp"])
def test_tiny_horizons(api, kind, N):
    """Horizon edge cases (the reference's scalar-integrator regressions use N in {1, 2, 4, 8}): the unrolled-by-two
    ping-pong loops, the LDS ring and the cooperative sweep must handle N smaller than their pipeline depth."""
    if kind == "scalar_path":
        p = api.scalar_integrator_problem(N, path_constraint=True)
    elif kind == "cartpole_box":
        p = api.cartpole_problem(api.SOLVER_IPDDP, True, N)
    else:
        p = api.pendulum_problem(api.SOLVER_CLDDP, True, N)
    B = 5
    x0 = api.batch_x0(p, B, 20261002, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0); hs.solve()
    res = hs.results(); X, U = hs.trajectory(); K, k = hs.gains()
    for b in range(B):
        o = api.Oracle(p); o.set_initial(x0[b]); r = o.solve()
        assert r["iterations"] == res["iterations"][b] and r["status"] == res["status"][b], (kind, N, b)
        conv = r["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE)
        # solves that stop on the iteration cap amplify last-bit sin / cos differences over 80 iterations
        assert rel_err(res["final_objective"][b], r["final_objective"]) < (TOL if conv else 1e-6)
        Xo, Uo = o.trajectory(); Ko, ko = o.gains()
        assert rel_err(U[b], Uo) < (1e-7 if conv else 1e-4) and rel_err(K[b], Ko) < (1e-6 if conv else 1e-3)
    hs.close()
