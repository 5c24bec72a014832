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
              "pendulum_term_eq", "path_term_eq", "term_eq_only", "cartpole_ipddp_box_state"]
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
    assert max(rel_err(K, tw.K_u), rel_err(k, tw.k_u), rel_err(Vx, tw.Vx), rel_err(Vxx, tw.Vxx)) < 1e-8
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


@pytest.mark.gpu
@pytest.mark.parametrize("name", STEP_CASES)
def test_hip_second_order_step_level(api, oracle_built, name):
    p = _problem(api, name)
    B = 6
    x0 = api.batch_x0(p, B, 20260928, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize()
    ok = hs.backward()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    alphas = api.Oracle(p).alphas()
    tr = hs.forward(alphas)
    hs.close()
    for b in range(B):
        o = api.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b]); o.initialize()
        assert o.backward(retry=True) == ok[b]
        Ko, ko = o.gains(); Vxo, Vxxo = o.value(); dVo, rego = o.backward_scalars()
        assert max(rel_err(K[b], Ko), rel_err(k[b], ko), rel_err(Vx[b], Vxo), rel_err(Vxx[b], Vxxo), rel_err(dV[b], dVo)) < TOL, (name, b)
        assert reg[b] == rego
        for a, alpha in enumerate(alphas):
            t = o.forward(alpha)
            assert tr[b, a]["success"] == t["success"], (name, b, alpha)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOLVE_CASES)
def test_hip_second_order_full_solve(api, oracle_built, name):
    p = _problem(api, name)
    B = 12
    x0 = api.batch_x0(p, B, 20260929, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
    res = hs.results(); X, U = hs.trajectory(); hs.close()
    ores, oX, oU, _, _ = api.oracle_solve_batch(p, x0, U0, n_threads=8)
    same = (res["iterations"] == ores["iterations"]) & (res["status"] == ores["status"])
    assert same.all(), list(zip(res["iterations"], ores["iterations"], res["status"], ores["status"]))
    conv = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    for b in range(B):
        if conv[b]:
            assert res["n_forward"][b] == ores["n_forward"][b] and rel_err(res["final_objective"][b], ores["final_objective"][b]) < 1e-7
            assert rel_err(X[b], oX[b]) < 1e-6 and rel_err(U[b], oU[b]) < 1e-6


@pytest.mark.gpu
def test_second_order_mode_is_refused_without_hessians(api):
    p = api.quadrotor_problem(api.SOLVER_IPDDP, 20, True)
    p.options.use_ilqr = 0
    with pytest.raises(api.HipError, match="Hessian"):
        api.HipBatchSolver(p, 4)
