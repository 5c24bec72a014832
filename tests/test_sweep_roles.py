"""Role-split IPDDP sweep (round 6): helper wavefronts inside the sweep workgroup evaluate what k_condense<.., true> (and k_post) evaluate and
hand the step records to the recursion wavefront through an LDS ring (kernels_coop.hpp::k_backward_ipddp_coop<.., NH > 0>).  The arithmetic is the
same device functions in the same order, so everything the sweep produces must be the SAME BITS as the separate kernels give
(CDDP_HIP_SWEEP_ROLES=0): linearisation stacks, gains, value function, slack / dual gains and step caps (through the trials of the first rollout),
whole solves incl. work counts -- for every path-constrained layout with nx <= 8, one to three helpers, batches that leave lanes and whole
workgroups idle, and problems whose factorisation fails and restarts the sweep with a larger regularisation (nu = 2)."""
import numpy as np
import pytest

from test_gpu_parity import make, spread_for

pytestmark = pytest.mark.gpu

ROLE_CASES = ["pendulum_ipddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_ipddp_box_state",
              "cartpole_ipddp_box_state", "unicycle_ipddp_box_state", "bicycle_ipddp_box", "car_ipddp_box", "hcw_ipddp_box",
              "unicycle_ipddp_box_soc", "unicycle_ipddp_thrust"]


def _run(api, p, B, x0, U0, alphas=(1.0, 0.5, 0.125)):
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize(); ok = hs.backward()
    K0, k0 = hs.gains(); Vx0, Vxx0 = hs.value(); dV0, reg0 = hs.backward_scalars()
    A0, B0 = hs.linearization()
    tr = hs.forward(list(alphas))      # trial records: alpha_pr / alpha_du carry the step caps, cost / merit / theta the slack and dual gains
    hs.solve()
    r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); Vx, Vxx = hs.value(); S, Y = hs.duals()[:2]; hs.close()
    out = [ok, K0, k0, Vx0, Vxx0, dV0, reg0, A0, B0, r["iterations"], r["status"], r["final_objective"], r["n_backward"], r["n_forward"], X, U, K, k, Vx, Vxx, S, Y]
    out += [tr[name] for name in tr.dtype.names]
    return out


@pytest.mark.parametrize("case", ROLE_CASES)
def test_role_split_sweep_agrees_bitwise(api, case, monkeypatch):
    p = make(api, case)
    B = 64 + 16 + 3          # one full tile, one full workgroup of the next tile and a partial one (idle lanes, idle workgroups in the XCD super-group)
    x0 = api.batch_x0(p, B, 20261101, spread_for(p))
    U0 = api.batch_U0(p, B)
    monkeypatch.setenv("CDDP_HIP_SWEEP_ROLES", "0")
    ref = _run(api, p, B, x0, U0)
    for nh in ("1", "2"):
        monkeypatch.setenv("CDDP_HIP_SWEEP_ROLES", nh)
        got = _run(api, p, B, x0, U0)
        assert len(ref) == len(got)
        for i, (a, g) in enumerate(zip(ref, got)):
            assert np.array_equal(a, g, equal_nan=True), (case, nh, i)


def test_role_split_sweep_restarts_with_larger_regularisation(api, monkeypatch):
    """nu = 2: a 2 x 2 factorisation that fails (here: rows poisoned with a NaN initial state, in the middle of lane groups of two different
    workgroups) makes the recursion wave ask the helpers for the step records of the whole horizon again, once per regularisation step
    (s_verdict), while the other trajectories of the wavefront wait; same restarts, same bits, and the clean rows are untouched."""
    p = make(api, "unicycle_ipddp_box_ball")
    p.options.max_iterations = 30
    B = 96
    x0 = api.batch_x0(p, B, 20261102, spread_for(p))
    U0 = api.batch_U0(p, B)
    x0[5, 1] = np.nan; x0[41, 0] = np.nan
    res = {}
    for nh in ("0", "1", "2"):
        monkeypatch.setenv("CDDP_HIP_SWEEP_ROLES", nh)
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); hs.close()
        res[nh] = (r["iterations"], r["status"], r["n_backward"], r["n_forward"], r["final_objective"], X, U, K, k)
    for nh in ("1", "2"):
        for a, g in zip(res["0"], res[nh]):
            assert np.array_equal(a, g, equal_nan=True), nh
    it, nb = res["0"][0], res["0"][2]
    assert int(nb[5]) > int(it[5]) + 1 and int(nb[41]) > int(it[41]) + 1, (nb[5], it[5])   # the poisoned rows did repeat sweeps


def test_role_split_sweep_full_batch_two_groups(api, monkeypatch):
    """The benchmark shape: 4096 cart-pole trajectories run as two tile groups on CU-masked streams, one sweep workgroup per CU."""
    p = api.cartpole_problem(api.SOLVER_IPDDP, True)
    B = 4096
    x0 = api.batch_x0(p, B, 20260928, [0.1, 0.3, 0.1, 0.1])
    out = {}
    for nh in ("0", "2"):
        monkeypatch.setenv("CDDP_HIP_SWEEP_ROLES", nh)
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0); st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); hs.close()
        out[nh] = (r["iterations"], r["status"], r["n_backward"], r["n_forward"], r["final_objective"], X, U)
        print("roles=%s solve_ms=%.2f" % (nh, st.solve_ms))
    for a, g in zip(out["0"], out["2"]):
        assert np.array_equal(a, g, equal_nan=True)
