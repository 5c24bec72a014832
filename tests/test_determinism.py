"""Repeated solves of one handle return the same bits: the atomics of the path (step caps by atomic min, max |Q_u| by
atomic max, the last-arriving-step logic of k_te_post, the ladder-shape histogram) are order-independent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["cartpole_box", "unicycle_box_ball", "manip7_terminal_eq"])
def test_repeated_solves_are_bitwise_identical(api, kind):
    if kind == "cartpole_box":
        p, B, spread = api.cartpole_problem(api.SOLVER_IPDDP, True), 320, np.array([0.3, 0.3, 0.1, 0.1])
    elif kind == "unicycle_box_ball":
        p, B, spread = api.unicycle_problem(api.SOLVER_IPDDP, 100, True), 320, 0.05 * np.ones(3)
    else:
        p, B, spread = api.manipulator7_problem(api.SOLVER_IPDDP, 20, terminal_equality=True, n_alphas=16), 96, 0.02 * np.ones(14)
    x0 = api.batch_x0(p, B, 20261013, spread)
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0, api.batch_U0(p, B))
    ref = None
    for _ in range(3):
        hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains()
        cur = (r["final_objective"].copy(), r["iterations"].copy(), r["status"].copy(), X.copy(), U.copy(), K.copy(), k.copy())
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert np.array_equal(a, b)
    hs.close()
