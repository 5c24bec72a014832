"""Full DDP (options.use_ilqr = false): the second-order dynamics terms of IPDDPSolver::backwardPass
(ipddp_solver.cpp:1070-1082 unconstrained, 1396-1408 path-constrained, 1160-1178 terminal equality with the costate as
value-gradient proxy; Hessian stacks cddp_solver_base.cpp:346-356; Hessian sources per plant: pendulum.cpp:68-85,
unicycle.cpp:68-89 + autodiff cross terms, cartpole.cpp:191-199 -> dynamical_system.cpp:137-217, lti_system.cpp:94-115).

CPU: the C++ oracle against the numpy twin (independent restatements) and the Hessians against finite differences of
the Jacobians; the reference's own pin (tests/cddp_core/test_ipddp_solver.cpp:1512-1578: use_ilqr = false changes k[0]
by > 1e-8).  GPU: the HIP path against the oracle, step level 1e-8 and solve level; unsupported plants are refused."""
import os
import sys

import numpy as np
import pytest

from test_gpu_parity import TERM_CASES, TOL, make, rel_err, spread_for

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

STEP_CASES = ["pendulum_ipddp_unc", "pendulum_ipddp_box", "cartpole_ipddp_unc", "cartpole_ipddp_box", "unicycle_ipddp_box", "unicycle_ipddp_box_ball",
              "pendulum_term_eq", "path_term_eq", "term_eq_only", "cartpole_ipddp_box_state",
              "bicycle_ipddp_box", "car_ipddp_box"]     # f3 tail: analytic + autodiff-default Hessians (bicycle), dual2nd of the discrete map (car)
SOLVE_CASES = ["pendulum_ipddp_unc", "pendulum_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_term_eq", "path_term_eq"]   # (cart-pole full DDP from
# the hanging start factors indefinite Q_uu blocks: its iterates are chaotic in the rounding, compared at step level only)


def _problem(api, name):
    p = TERM_CASES[name](api) if name in TERM_CASES else make(api, name)
    p.options.use_ilqr = 0
    return p


def _twin(name):
    import make_twin_golden as G
    spec = G.CASES[name](); spec["options"]["use_ilqr"] = False
    tw = G.T.Twin(spec)
    tw.set_initial(np.array(spec["x0"], float), spec.get("U0"))
    return tw


@pytest.mark.parametrize("name", ["pendulum_ipddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box_ball"])
def test_hessians_match_finite_differences_of_the_jacobians(api, oracle_built, name):
    p = make(api, name)
    o = api.Oracle(p)
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, p.nx); u = rng.uniform(-1, 1, p.nu)
    Fxx, Fuu, Fux = o.hessians(x, u)
    h = 1e-6
    for j in range(p.nx):
        e = np.zeros(p.nx); e[j] = h
        _, _, Ap, Bp = o.dynamics(x + e, u); _, _, Am, Bm = o.dynamics(x - e, u)
        dA = (Ap - Am) / (2 * h); dB = (Bp - Bm) / (2 * h)          # d f_x / d x_j, d f_u / d x_j
        assert np.max(np.abs(Fxx[:, :, j] - dA)) < 1e-6, (name, j)
        if name != "pendulum_ipddp_box":    # Pendulum's cross Hessian comes from its -sin autodiff twin (zero either way)
            assert np.max(np.abs(Fux[:, :, j] - np.swapaxes(dB, 0, 1)[:, :, None].reshape(p.nx, p.nu) if False else Fux[:, :, j] - dB)) < 1e-6
    for j in range(p.nu):
        e = np.zeros(p.nu); e[j] = h
        _, _, _, Bp = o.dynamics(x, u + e); _, _, _, Bm = o.dynamics(x, u - e)
        assert np.max(np.abs(Fuu[:, :, j] - (Bp - Bm) / (2 * h))) < 1e-6


@pytest.mark.parametrize("name", [c for c in STEP_CASES if c != "cartpole_ipddp_box_state"])
def test_oracle_matches_twin_with_second_order_terms(api, oracle_built, name):
    tw = _twin(name); tw.initialize(); tw.X_lin, tw.U_lin = tw.X, tw.U
    okt = tw.backward()
    p = _problem(api, name)
    U0 = api.batch_U0(p, 1)
    o = api.Oracle(p); o.set_initial(p.x0, None if U0 is None else U0[0]); o.initialize()
    assert bool(o.backward(retry=False)) == bool(okt)
    K, k = o.gains(); Vx, Vxx = o.value()
    # (1e-8: with the tensor terms Q_uu may be indefinite -- LDLT accepts it -- and the gains reach 1e3..1e4)
    # (bicycle / car: gains of 1e3..6e3 from the initial guess; the oracle against ITSELF with <= 1 ulp noise on its sines moves by 6e-9..8e-9 there)
    assert max(rel_err(K, tw.K_u), rel_err(k, tw.k_u), rel_err(Vx, tw.Vx), rel_err(Vxx, tw.Vxx)) < (1e-7 if name in ("bicycle_ipddp_box", "car_ipddp_box") else 1e-8)
    # the second-order terms are not a no-op: the feed-forward gain moves by more than 1e-8 (test_ipddp_solver.cpp:1512-1578)
    p2 = TERM_CASES[name](api) if name in TERM_CASES else make(api, name)
    o2 = api.Oracle(p2); o2.set_initial(p2.x0, None if U0 is None else U0[0]); o2.initialize(); o2.backward(retry=False)
    if name.startswith("unicycle"):   # (the pendulum / cart-pole examples start AT an equilibrium with zero controls, where every
        assert np.max(np.abs(o2.gains()[1] - k)) > 1e-8   # second derivative that enters vanishes; LTI plants have none)
    if name in SOLVE_CASES:
        tw2 = _twin(name); r = tw2.solve()
        o3 = api.Oracle(p); o3.set_initial(p.x0, None if U0 is None else U0[0]); ro = o3.solve()
        assert (ro["iterations"], ro["status"], ro["n_backward"], ro["n_forward"]) == (r["iterations"], r["status"], r["n_backward"], r["n_forward"])
        assert rel_err(ro["final_objective"], r["final_objective"]) < 1e-7


# With the tensor terms Q_uu is no longer a sum of PSD pieces: from a random initial guess it is often indefinite (LDLT accepts it)
# and the gains reach 1e4..1e10.  Such a sweep amplifies a last-bit difference of sin / cos by many orders of magnitude -- the
# oracle against ITSELF with <= 1 ulp noise on its trig results (oracle/models.hpp::trig_noise) moves by up to O(1) there.  Each
# trajectory is therefore compared at max(1e-8, 1e3 x its own noise yardstick); cart-pole cases start 0.1 x the usual spread
# from the hanging equilibrium, where every trajectory is well conditioned (yardstick <= 1e-10) and the strict 1e-8 bar applies.
STEP_SCALE = {"cartpole_ipddp_unc": 0.1, "cartpole_ipddp_box": 0.1, "cartpole_ipddp_box_state": 0.1}
STRICT_EVERYWHERE = ["pendulum_ipddp_unc", "pendulum_ipddp_box", "cartpole_ipddp_unc", "cartpole_ipddp_box", "cartpole_ipddp_box_state",
                     "pendulum_term_eq", "path_term_eq", "term_eq_only"]


def _oracle_lib(api):
    import ctypes
    return ctypes.CDLL(api.ORACLE_LIB_PATH)


def _oracle_sweep(api, p, x0b, U0b):
    o = api.Oracle(p); o.set_initial(x0b, U0b); o.initialize()
    ok = o.backward(retry=True)
    K, k = o.gains(); Vx, Vxx = o.value(); dV, reg = o.backward_scalars()
    return o, ok, (K, k, Vx, Vxx, dV), reg


def _noise_yardstick(api, lib, p, x0b, U0b, ref_ok, ref, ref_reg):
    worst = 0.0
    try:
        for _ in range(2):
            lib.cddp_oracle_set_trig_noise(1)
            _, ok, got, reg = _oracle_sweep(api, p, x0b, U0b)
            if ok != ref_ok or reg != ref_reg: return np.inf
            worst = max(worst, max(rel_err(g, r) for g, r in zip(got, ref)))
    finally:
        lib.cddp_oracle_set_trig_noise(0)
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", STEP_CASES)
def test_hip_second_order_step_level(api, oracle_built, name):
    """One sweep + the ladder with the tensor terms, HIP parity build (shared sin / cos, tests/test_shared_trig_parity.py) against the
    oracle in its shared-trig mode: with the same trig routine on both sides there is no libm noise left to amplify, so EVERY
    trajectory is held to the strict 1e-8 -- also the unicycle sweeps whose indefinite Q_uu drives the gains to 1e4..1e10 (round 2
    compared those at 1e3 x a noise yardstick, and its `strict >= 0` clause was vacuous)."""
    p = _problem(api, name)
    B = 6
    spread = np.asarray(spread_for(p)) * STEP_SCALE.get(name, 1.0) if p.nx > 1 else 0.05 * np.ones(1)
    x0 = api.batch_x0(p, B, 20260928, spread)
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B, trig="shared"); hs.set_initial(x0, U0); hs.initialize()
    ok = hs.backward()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    alphas = api.Oracle(p).alphas()
    tr = hs.forward(alphas)
    hs.close()
    worst = 0.0
    with api.shared_trig():
        for b in range(B):
            U0b = None if U0 is None else U0[b]
            o, oko, ref, rego = _oracle_sweep(api, p, x0[b], U0b)
            err = max(rel_err(g, r) for g, r in zip((K[b], k[b], Vx[b], Vxx[b], dV[b]), ref))
            worst = max(worst, err)
            assert oko == ok[b] and reg[b] == rego, (name, b)
            assert err < TOL, (name, b, err)
            for a, alpha in enumerate(alphas):
                t = o.forward(alpha)
                assert tr[b, a]["success"] == t["success"], (name, b, alpha)
    print("%s: worst HIP-vs-oracle relative error %.2e over %d trajectories (shared-trig mode)" % (name, worst, B))


@pytest.mark.gpu
@pytest.mark.parametrize("name", STEP_CASES)
def test_hip_second_order_step_level_default_build(api, oracle_built, name):
    """The product build (device libm) against the glibc-mode oracle: each trajectory at max(1e-8, 1e3 x its own noise yardstick)
    -- the amplification of a <= 1 ulp libm difference by an indefinite Q_uu is a property of the problem, not of the port; the
    well-conditioned cases must meet the strict bar everywhere."""
    p = _problem(api, name)
    lib = _oracle_lib(api)
    B = 6
    spread = np.asarray(spread_for(p)) * STEP_SCALE.get(name, 1.0) if p.nx > 1 else 0.05 * np.ones(1)
    x0 = api.batch_x0(p, B, 20260928, spread)
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize()
    ok = hs.backward()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    hs.close()
    strict = 0
    for b in range(B):
        U0b = None if U0 is None else U0[b]
        o, oko, ref, rego = _oracle_sweep(api, p, x0[b], U0b)
        yard = _noise_yardstick(api, lib, p, x0[b], U0b, oko, ref, rego)
        err = max(rel_err(g, r) for g, r in zip((K[b], k[b], Vx[b], Vxx[b], dV[b]), ref))
        print("%s traj %d: HIP-vs-oracle %.2e, oracle-vs-noisy-oracle %.2e" % (name, b, err, yard))
        if yard <= 1e-10:
            strict += 1
            assert oko == ok[b] and reg[b] == rego
            assert err < TOL, (name, b, err)
        elif np.isfinite(yard):
            assert oko == ok[b] and reg[b] == rego
            assert err < max(TOL, 1e3 * yard), (name, b, err, yard)
    if name in STRICT_EVERYWHERE:
        assert strict == B, (name, strict)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOLVE_CASES)
def test_hip_second_order_full_solve(api, oracle_built, name):
    """Full-DDP solves, parity build vs the oracle's shared-trig mode: every trajectory has the oracle's status, iteration count and
    rollout count (strict, no yardstick); converged ones agree in objective 1e-7 and trajectory 1e-6."""
    p = _problem(api, name)
    B = 12
    x0 = api.batch_x0(p, B, 20260929, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B, trig="shared"); hs.set_initial(x0, U0); hs.solve()
    res = hs.results(); X, U = hs.trajectory(); hs.close()
    with api.shared_trig():
        ores, oX, oU, _, _ = api.oracle_solve_batch(p, x0, U0, n_threads=8)
    same = (res["iterations"] == ores["iterations"]) & (res["status"] == ores["status"]) & (res["n_forward"] == ores["n_forward"])
    print("%s: HIP == oracle on %d / %d trajectories (shared-trig mode)" % (name, same.sum(), B))
    assert same.all(), list(zip(res["iterations"], ores["iterations"], res["status"], ores["status"]))
    conv = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    for b in range(B):
        if conv[b]:
            assert rel_err(res["final_objective"][b], ores["final_objective"][b]) < 1e-7
            assert rel_err(X[b], oX[b]) < 1e-6 and rel_err(U[b], oU[b]) < 1e-6


@pytest.mark.gpu
def test_second_order_mode_is_accepted_for_every_builtin_plant(api):
    """Rounds 2-3 refused use_ilqr = 0 for the quadrotor and the two synthetic plants (no device Hessians); round 4 lifted that."""
    for p in (api.quadrotor_problem(api.SOLVER_IPDDP, 20, True), api.quadrotor12_problem(api.SOLVER_IPDDP, 20, True), api.manipulator7_problem(api.SOLVER_IPDDP, 10)):
        p.options.use_ilqr = 0
        hs = api.HipBatchSolver(p, 4)
        hs.close()


# ------------------------------------------------------------------------------------------------------------------
# Round 4: full DDP on the large plants (VERDICT r03 missing #3).  The quadrotor's dual2nd frame (17 seeds, 307 doubles per dual
# number) does not fit a GPU lane; the device assembles the second-order terms from 8-seed evaluations, one per pair of 4-variable
# blocks (dev_models.hpp::ad_tensor_terms_blocked) -- component-wise the same arithmetic as the full evaluation.  Also the two
# synthetic plants of BASELINE configs [3] / [4] (no Hessian overrides: the base class's dual2nd default on the plant's expression).
# ------------------------------------------------------------------------------------------------------------------
BIG_DDP_CASES = ["quadrotor_ipddp_box", "quad12_ipddp_box", "manip7_ipddp_box", "manip7_term_eq_parallel_ls"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", BIG_DDP_CASES)
def test_hip_second_order_large_plants_step_level(api, oracle_built, name):
    p = _problem(api, name)
    B = 3
    x0 = api.batch_x0(p, B, 20261104, spread_for(p))
    U0 = api.batch_U0(p, B)
    X0 = np.tile(p.X0_single, (B, 1, 1)) if hasattr(p, "X0_single") else None
    if X0 is not None:
        X0[:, 0, :] = x0
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0, X0); hs.initialize()
    ok = hs.backward()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    hs.close()
    p1 = TERM_CASES[name](api) if name in TERM_CASES else make(api, name)     # Gauss-Newton twin of the same problem
    worst = 0.0; moved = 0.0
    for b in range(B):
        o = api.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b], None if X0 is None else X0[b]); o.initialize()
        oko = o.backward(retry=True)
        Ko, ko = o.gains(); Vxo, Vxxo = o.value(); dVo, rego = o.backward_scalars()
        assert oko == ok[b] and reg[b] == rego, (name, b, reg[b], rego)
        err = max(rel_err(g, r) for g, r in zip((K[b], k[b], Vx[b], Vxx[b], dV[b]), (Ko, ko, Vxo, Vxxo, dVo)))
        worst = max(worst, err)
        assert err < TOL, (name, b, err)
        o1 = api.Oracle(p1); o1.set_initial(x0[b], None if U0 is None else U0[b], None if X0 is None else X0[b]); o1.initialize(); o1.backward(retry=True)
        moved = max(moved, float(np.max(np.abs(o1.gains()[1] - ko))))
    if name not in TERM_CASES:   # (terminal-equality branch: the costate iterate is the value-gradient proxy, zero at the first sweep, :1160-1178)
        assert moved > 1e-8, "the second-order terms changed nothing (test_ipddp_solver.cpp:1512-1578 asks for > 1e-8 in k)"
    print("%s full DDP: worst HIP-vs-oracle relative error %.2e, k moves by %.2e against Gauss-Newton" % (name, worst, moved))
