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


@pytest.mark.parametrize("kind", ["cartpole_box", "unicycle_box_ball", "cartpole_clddp"])
def test_tile_groups_do_not_change_results(api, kind, monkeypatch):
    """cddp_hip_create cuts the batch into tile groups solved concurrently on separate streams (CDDP_HIP_GROUPS pins the
    count): iterates, gains, counters and the gather records must be the same bits for 1, 2, 3 and 5 groups."""
    if kind == "cartpole_box":
        p, B, spread = api.cartpole_problem(api.SOLVER_IPDDP, True), 1000, np.array([0.1, 0.3, 0.1, 0.1])
    elif kind == "unicycle_box_ball":
        p, B, spread = api.unicycle_problem(api.SOLVER_IPDDP, 100, True), 700, 0.05 * np.ones(3)
    else:
        p, B, spread = api.cartpole_problem(api.SOLVER_CLDDP, True), 1000, np.array([0.1, 0.3, 0.1, 0.1])
    x0 = api.batch_x0(p, B, 20261014, spread)
    U0 = api.batch_U0(p, B)
    ref = None
    for ng in (1, 2, 3, 5):
        monkeypatch.setenv("CDDP_HIP_GROUPS", str(ng))
        hs = api.HipBatchSolver(p, B)
        assert hs.num_groups() == ng
        hs.set_initial(x0, U0)
        st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); Vx, Vxx = hs.value()
        S, Y, G = hs.duals() if p.c.solver == api.SOLVER_IPDDP else (None, None, None)
        hs.close()
        cur = [r[f].copy() for f in r.dtype.names] + [X, U, K, k, Vx, Vxx] + ([S, Y, G] if S is not None else [])
        cur.append(np.array([st.sweeps, st.rollouts, st.traj_iterations, st.rollout_steps, st.n_converged]))
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert np.array_equal(a, b), (kind, ng)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cartpole_ipddp", "cartpole_clddp", "unicycle_ipddp"])
def test_graph_replay_of_iteration_windows_is_bitwise_the_stream_loop(api, kind, monkeypatch):
    """CDDP_HIP_GRAPH=1 (experiment): the iterations between two polls captured into a hipGraph per (ladder shape, window length,
    last flag) and replayed -- same kernels, same arguments, so every result word is the same."""
    p = {"cartpole_ipddp": lambda: api.cartpole_problem(api.SOLVER_IPDDP, True), "cartpole_clddp": lambda: api.cartpole_problem(api.SOLVER_CLDDP, True),
         "unicycle_ipddp": lambda: api.unicycle_problem(api.SOLVER_IPDDP, 100, True)}[kind]()
    B = 200
    x0 = api.batch_x0(p, B, 20270201, 0.05 * np.ones(p.nx))

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0); st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); hs.close()
        return [r[f].copy() for f in r.dtype.names] + [X, U, K, k, np.array([st.sweeps, st.rollouts, st.outer_iterations])]

    monkeypatch.delenv("CDDP_HIP_GRAPH", raising=False)
    ref = run()
    monkeypatch.setenv("CDDP_HIP_GRAPH", "1")
    got = run(); got2 = run()
    for a, b, c in zip(ref, got, got2):
        assert np.array_equal(a, b) and np.array_equal(a, c), kind


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cartpole_ipddp", "cartpole_logddp"])
def test_successive_chunks_of_a_large_batch_do_not_change_results(api, kind, monkeypatch):
    """A batch above 8192 trajectories is solved as successive chunks on the handle (capi.hip::pick_groups, one chunk in flight at a
    time).  CDDP_HIP_CHUNK sets the chunk size: 1000 trajectories in chunks of 256 (4 chunks, the last one
    short) against one group -- every result word, trajectory and gain the same bits, and the work counters add up."""
    p = api.cartpole_problem(api.SOLVER_IPDDP if kind == "cartpole_ipddp" else api.SOLVER_LOGDDP, True)
    B = 1000
    x0 = api.batch_x0(p, B, 20270202, np.array([0.1, 0.3, 0.1, 0.1]))
    monkeypatch.delenv("CDDP_HIP_GROUPS", raising=False)

    def run(chunk, ng):
        monkeypatch.setenv("CDDP_HIP_CHUNK", chunk)
        hs = api.HipBatchSolver(p, B)
        assert hs.num_groups() == ng
        hs.set_initial(x0); st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); hs.close()
        return [r[f].copy() for f in r.dtype.names] + [X, U, K, k, np.array([st.sweeps, st.rollouts, st.traj_iterations, st.rollout_steps, st.n_converged])]

    ref = run("0", 1)
    got = run("256", 4)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b), kind


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cartpole_ipddp", "cartpole_clddp", "unicycle_ipddp"])
@pytest.mark.parametrize("plan", ["F=0-192,W=192-256", "C=128-256,F=0-128", "F=0-176,W=176-208|F=0-176,W=208-240"])
def test_cu_partitioned_streams_do_not_change_results(api, kind, plan, monkeypatch):
    """CDDP_HIP_CUMASK (round 5): the rollout and the serial sweep of a tile group on streams created with
    hipExtStreamCreateWithCUMask, ordered against the group's main stream by events -- same kernels, same arguments, same
    order per trajectory (cddp_solver_base.cpp:29-186), so every result word is the one-stream solve's; also with two groups
    whose rollouts alternate (CDDP_HIP_PINGPONG)."""
    p = {"cartpole_ipddp": lambda: api.cartpole_problem(api.SOLVER_IPDDP, True), "cartpole_clddp": lambda: api.cartpole_problem(api.SOLVER_CLDDP, True),
         "unicycle_ipddp": lambda: api.unicycle_problem(api.SOLVER_IPDDP, 100, True)}[kind]()
    B = 600
    x0 = api.batch_x0(p, B, 20270301, 0.05 * np.ones(p.nx))

    def run():
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0); st = hs.solve()
        r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains(); hs.close()
        return [r[f].copy() for f in r.dtype.names] + [X, U, K, k, np.array([st.sweeps, st.rollouts, st.traj_iterations, st.rollout_steps])]

    for v in ("CDDP_HIP_CUMASK", "CDDP_HIP_GROUPS", "CDDP_HIP_PINGPONG"):
        monkeypatch.delenv(v, raising=False)
    ref = run()
    monkeypatch.setenv("CDDP_HIP_CUMASK", plan)
    for groups, pp in ((1, 0), (2, 0), (2, 1)):
        monkeypatch.setenv("CDDP_HIP_GROUPS", str(groups))
        monkeypatch.setenv("CDDP_HIP_PINGPONG", str(pp))
        cur = run()
        for a, b in zip(ref, cur):
            assert np.array_equal(a, b), (kind, plan, groups, pp)
