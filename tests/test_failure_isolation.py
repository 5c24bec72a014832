"""Failure isolation: a poisoned trajectory does not touch its neighbours.

The reference treats a non-finite trial as a failed line-search step (ipddp_solver.cpp:1615-1656: `!x.allFinite()` -> trial
rejected; :1778-1782: a forward pass without an accepted trial raises the regularisation until the limit ends the solve), one
problem at a time.  In the batched solver 64 trajectories share a wavefront, 4 share the lane group of a cooperative sweep, and
the line-search launch shape follows batch-wide statistics -- so the property to check is that a NaN initial state and an Inf
initial control INSIDE a 256-trajectory batch (a) end with the failure status the oracle gives the same poisoned problem and
(b) leave every other trajectory's results bit for bit what they are in the clean batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAILED = None


def _solve(api, p, x0, U0):
    hs = api.HipBatchSolver(p, x0.shape[0])
    hs.set_initial(x0, U0)
    hs.solve()
    r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains()
    hs.close()
    return r, X, U, K, k


@pytest.mark.parametrize("solver", ["ipddp", "clddp", "logddp", "msipddp"])
def test_nan_and_inf_rows_do_not_poison_the_batch(api, oracle_built, solver):
    # (the resident LogDDP / MSIPDDP kernels, round 4: one trajectory per lane, the same batch-wide ladder statistics -- isolation is
    #  asserted for them too; MSIPDDP on the pendulum, where its clean batch converges)
    if solver == "msipddp":
        p = api.pendulum_problem(api.SOLVER_MSIPDDP, True); p.c.solver = api.SOLVER_MSIPDDP
    else:
        p = api.cartpole_problem({"ipddp": api.SOLVER_IPDDP, "clddp": api.SOLVER_CLDDP, "logddp": api.SOLVER_LOGDDP}[solver], True)
    p.options.max_iterations = 40
    B = 256
    x0 = api.batch_x0(p, B, 20261101, [0.1, 0.3, 0.1, 0.1][:p.nx])
    U0 = api.batch_U0(p, B)
    if U0 is None:
        U0 = np.zeros((B, p.N, p.nu))
    clean = _solve(api, p, x0, U0)
    bad_x, bad_u = 77, 130            # two different wavefronts, each in the middle of a 4-trajectory lane group
    x0p = x0.copy(); U0p = U0.copy()
    x0p[bad_x, 1] = np.nan
    U0p[bad_u, 17, 0] = np.inf
    pois = _solve(api, p, x0p, U0p)
    keep = np.ones(B, dtype=bool); keep[[bad_x, bad_u]] = False
    for name in clean[0].dtype.names:
        assert np.array_equal(clean[0][name][keep], pois[0][name][keep]), name
    for a, b in zip(clean[1:], pois[1:]):
        assert np.array_equal(a[keep], b[keep])
    # The poisoned rows end as the reference's logic ends them -- the oracle solves the same two poisoned problems alone.  IPDDP: a
    # non-finite trial is a failed step (ipddp_solver.cpp:1615-1656), every step size fails, the regularisation climbs to its limit
    # (:1778-1782): a failure status.  CLDDP has no such guard: max / lpNorm drop the NaN (std::max(a, NaN) = a,
    # clddp_solver.cpp:193-213) and the solver REPORTS convergence after one iteration on a NaN state -- the reference's behaviour,
    # restated by the oracle and reproduced here (status 1 after 1 iteration on both sides), not a property to "fix" in a drop-in;
    # with an infinite control the two sides part ways inside inf - inf arithmetic (the status of such a row means nothing in the
    # reference either), so for CLDDP only the isolation and the non-finite objective are asserted.
    if solver == "ipddp":
        ores = api.oracle_solve_batch(p, x0p[[bad_x, bad_u]], U0p[[bad_x, bad_u]], n_threads=2, want_traj=False)[0]
        for j, b in enumerate((bad_x, bad_u)):
            assert int(pois[0]["status"][b]) == int(ores["status"][j]) and int(pois[0]["iterations"][b]) == int(ores["iterations"][j]), (b, pois[0][b], ores[j])
            assert int(pois[0]["status"][b]) in (api.STATUS_MAX_ITERATIONS, api.STATUS_REG_LIMIT), (b, pois[0][b])
    # a poisoned row is recognisable: its objective is not a finite number, or it ended with a failure status
    for b in (bad_x, bad_u):
        assert (not np.isfinite(pois[0]["final_objective"][b])) or int(pois[0]["status"][b]) in (api.STATUS_MAX_ITERATIONS, api.STATUS_REG_LIMIT), (b, pois[0][b])
        if solver in ("ipddp", "clddp"):
            assert not np.isfinite(pois[0]["final_objective"][b]), (b, pois[0][b])
    # the clean batch itself is not all failures: the comparison above is not vacuous
    assert np.all(np.isfinite(clean[0]["final_objective"]))
