"""Stack-fed mode (`cddp_hip_backward_stacks`, include/cddp_hip.h): the host evaluates the plug-ins and hands
over the (N x batch) derivative stacks; the GPU runs the Riccati sweep on them.  Checked against the oracle's
gains and value function on every trajectory, and for consistency against the device solver's own sweep on the same iterate.  Tolerance 1e-9 relative: the stacks are assembled in numpy, not in the kernels'
association order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def build_stacks(api, p, X, U):
    """Host-side plug-in evaluation: A = I + dt f_x, B = dt f_u from the oracle's model, quadratic objective."""
    B, N = U.shape[0], U.shape[1]
    nx, nu, dt = p.nx, p.nu, p.dt
    o = api.Oracle(p)
    fx = np.zeros((B, N, nx, nx)); fu = np.zeros((B, N, nx, nu))
    for b in range(B):
        for t in range(N):
            _, _, Fx, Fu = o.dynamics(X[b, t], U[b, t])
            fx[b, t] = dt * Fx + np.eye(nx); fu[b, t] = dt * Fu
    Qd, Rd = p.Q * dt, p.R * dt
    lx = np.einsum("ij,btj->bti", 2.0 * Qd, X[:, :N] - p.x_ref[None, None, :])
    lu = np.einsum("ij,btj->bti", 2.0 * Rd, U)
    lxx = np.broadcast_to(2.0 * Qd, (B, N, nx, nx)).copy()
    luu = np.broadcast_to(2.0 * Rd, (B, N, nu, nu)).copy()
    lux = np.zeros((B, N, nu, nx))
    VxN = np.einsum("ij,bj->bi", 2.0 * p.Qf, X[:, N] - p.x_ref[None, :])
    H = 2.0 * p.Qf
    VxxN = np.broadcast_to(0.5 * (H + H.T), (B, nx, nx)).copy()
    return fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN


@pytest.mark.parametrize("case", ["pendulum_ipddp", "cartpole_ipddp", "cartpole_clddp", "unicycle_clddp"])
def test_stack_fed_sweep_matches_solver_sweep(api, case):
    S = api
    if case == "pendulum_ipddp":
        p = S.pendulum_problem(S.SOLVER_IPDDP, False); reg_in_value = 1
    elif case == "cartpole_ipddp":
        p = S.cartpole_problem(S.SOLVER_IPDDP, False); reg_in_value = 1
    elif case == "cartpole_clddp":
        p = S.cartpole_problem(S.SOLVER_CLDDP, False); reg_in_value = 0
    else:
        p = S.unicycle_problem(S.SOLVER_CLDDP, 60, False); p._cons = []; p._rebuild(); reg_in_value = 0
    B = 6
    x0 = S.batch_x0(p, B, 20260930)
    U0 = S.batch_U0(p, B) if hasattr(S, "batch_U0") else None
    hs = S.HipBatchSolver(p, B)
    hs.set_initial(x0, U0); hs.initialize()
    ok = hs.backward()
    assert ok.all()
    X, U = hs.trajectory()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    hs.close()
    assert np.all(reg == reg[0])
    stacks = build_stacks(S, p, X, U)
    K2, k2, Vx2, Vxx2, dV2, ok2, ms = S.hip_backward_stacks(*stacks, reg=float(reg[0]), reg_in_value=reg_in_value)
    assert ok2.all()
    assert rel(K2, K) < TOL and rel(k2, k) < TOL
    assert rel(Vx2, Vx) < TOL and rel(Vxx2, Vxx) < TOL
    assert rel(dV2, dV) < TOL
    # and against the oracle directly, every trajectory of the batch (the comparison with the device solver's own sweep above is a
    # consistency check between two products of this repository; THIS is the parity check)
    for b in range(B):
        o = S.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b]); o.initialize()
        assert o.backward()
        Ko, ko = o.gains(); Vxo, Vxxo = o.value()
        assert rel(K2[b], Ko) < 1e-8 and rel(k2[b], ko) < 1e-8, b
        assert rel(Vx2[b], Vxo) < 1e-8 and rel(Vxx2[b], Vxxo) < 1e-8, b


def test_stack_fed_rejects_uninstantiated_shape(api):
    B, N, nx, nu = 2, 5, 5, 2   # (5, 2) is not an instantiated (nx, nu) pair
    z = lambda *s: np.zeros(s)
    with pytest.raises(api.HipError):
        api.hip_backward_stacks(z(B, N, nx, nx), z(B, N, nx, nu), z(B, N, nx), z(B, N, nu), z(B, N, nx, nx), z(B, N, nu, nu),
                                z(B, N, nu, nx), z(B, nx), z(B, nx, nx), reg=1e-6, reg_in_value=1)


# ----------------------------------------------------------------------------------------------------------------------
# Handle-bound, path-constrained stack-fed mode on USER-DEFINED plugins (round 2)
# ----------------------------------------------------------------------------------------------------------------------
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "twin"))


class QuadraticScalarSystem:
    """The reference's own user-defined DynamicalSystem (tests/cddp_core/test_ipddp_solver.cpp:291-346): a plant no
    built-in device model covers.  x+ = x + u + x^2 / 2; getStateJacobian returns 1 + x, getControlJacobian 1."""
    nx, nu = 1, 1
    discrete = True

    def step(self, x, u, t):
        return np.array([x[0] + u[0] + 0.5 * x[0] * x[0]])

    def jac(self, x, u, t):
        return np.array([[1.0 + x[0]]]), np.array([[1.0]])


class TiltedUnicycle:
    """A second host plugin (nx = 3, nu = 2): unicycle with a state-dependent speed gain -- no device counterpart."""
    nx, nu = 3, 2

    def f(self, x, u, t):
        g = 1.0 + 0.1 * np.tanh(x[0])
        return np.array([g * u[0] * np.cos(x[2]), g * u[0] * np.sin(x[2]), u[1]])

    def jac(self, x, u, t):
        g = 1.0 + 0.1 * np.tanh(x[0]); dg = 0.1 * (1.0 - np.tanh(x[0]) ** 2)
        A = np.zeros((3, 3)); B = np.zeros((3, 2))
        A[0, 0] = dg * u[0] * np.cos(x[2]); A[1, 0] = dg * u[0] * np.sin(x[2])
        A[0, 2] = -g * u[0] * np.sin(x[2]); A[1, 2] = g * u[0] * np.cos(x[2])
        B[0, 0] = g * np.cos(x[2]); B[1, 0] = g * np.sin(x[2]); B[2, 1] = 1.0
        return A, B


def _plugin_problem(kind):
    import cddp_twin as T
    if kind == "quadratic_scalar_box":
        return dict(solver="IPDDP", model=QuadraticScalarSystem(), integrator="euler", dt=1.0, N=6, Q=np.zeros((1, 1)), R=1e-2 * np.eye(1),
                    Qf=10.0 * np.eye(1), xref=[0.3], constraints={"ControlConstraint": T.ControlBox([-0.4], [0.4])},
                    options=dict(max_iterations=30, tolerance=1e-6, reg_initial_value=1e-6, mu_initial=0.1)), np.array([0.5])
    if kind == "quadratic_scalar_linear":
        return dict(solver="IPDDP", model=QuadraticScalarSystem(), integrator="euler", dt=1.0, N=6, Q=np.zeros((1, 1)), R=1e-2 * np.eye(1),
                    Qf=10.0 * np.eye(1), xref=[0.3], constraints={"StateUpper": T.Linear(np.eye(1), [0.9])},
                    options=dict(max_iterations=30, tolerance=1e-6, reg_initial_value=1e-6, mu_initial=0.1)), np.array([0.5])
    if kind == "tilted_unicycle_box_ball":
        return dict(solver="IPDDP", model=TiltedUnicycle(), integrator="rk4", dt=0.03, N=60, Q=np.zeros((3, 3)), R=0.05 * np.eye(2),
                    Qf=np.diag([100.0, 100.0, 50.0]), xref=[1.5, 1.5, np.pi / 2],
                    constraints={"control_limits": T.ControlBox([-1.1, -np.pi], [1.1, np.pi]), "obstacle": T.Ball(0.3, [0.8, 0.8])},
                    options=dict(max_iterations=40, tolerance=1e-4)), np.array([0.0, 0.0, np.pi / 4])
    raise KeyError(kind)


def _twin_stacks(tw):
    """What a host-plugin adapter hands to cddp_hip_set_stacks / cddp_hip_set_constraint_stacks for the twin's iterate."""
    N, nx, nu, m = tw.N, tw.nx, tw.nu, tw.m
    fx = np.zeros((N, nx, nx)); fu = np.zeros((N, nx, nu)); lx = np.zeros((N, nx)); lu = np.zeros((N, nu))
    Gx = np.zeros((N, m, nx)); Gu = np.zeros((N, m, nu))
    for t in range(N):
        fx[t], fu[t] = tw.lin(t)
        lx[t], lu[t], lxx, luu, lux = tw.cost_derivs(t)
        off = 0
        for _, c in tw.cons:
            gx, gu = c.jac(tw.X[t], tw.U[t]); Gx[t, off:off + c.dim] = gx; Gu[t, off:off + c.dim] = gu; off += c.dim
    H = 2.0 * tw.Qf
    return dict(fx=fx, fu=fu, lx=lx, lu=lu, lxx=np.tile(lxx, (N, 1, 1)), luu=np.tile(luu, (N, 1, 1)), lux=np.tile(lux, (N, 1, 1)),
                VxN=2.0 * tw.Qf @ (tw.X[N] - tw.xref), VxxN=H, y=getattr(tw, "Y", np.zeros((N, m))).copy(), s=getattr(tw, "S", np.zeros((N, m))).copy(),
                g=getattr(tw, "G", np.zeros((N, m))).copy(), Gx=Gx, Gu=Gu)


@pytest.mark.parametrize("kind", ["quadratic_scalar_box", "quadratic_scalar_linear", "tilted_unicycle_box_ball"])
def test_constrained_stack_fed_sweep_on_user_plugins(api, kind):
    """A batch of host-plugin problems through the handle-bound stack-fed mode: at the initial iterate AND at later
    iterates of the twin's own solve, the GPU sweep (K, k, V_x, V_xx, dV, k_y, K_y, k_s, K_s, dX, inf_*, step caps, the
    regularisation-retry loop) equals the numpy twin's backward pass on the same stacks (1e-9: numpy BLAS order vs
    the kernel's scalar loops)."""
    import cddp_twin as T
    spec, x0 = _plugin_problem(kind)
    rng = np.random.default_rng(20261020)
    B = 5
    twins = []
    for b in range(B):
        tw = T.Twin(dict(spec))
        xb = x0 + (0.0 if b == 0 else 1.0) * rng.uniform(-0.05, 0.05, size=x0.shape)
        U0 = np.tile([0.4, 0.1], (tw.N, 1)) if tw.nu == 2 else None
        tw.set_initial(xb, U0); tw.initialize(); tw.X_lin, tw.U_lin = tw.X, tw.U
        twins.append(tw)
    tw0 = twins[0]
    hs = api.HipStackSolver(B, tw0.nx, tw0.nu, tw0.m, tw0.N)
    opt = api.default_options()
    for outer in range(4):          # iterate 0 and three accepted iterates later
        st = [_twin_stacks(tw) for tw in twins]
        stack = lambda key: np.stack([s_[key] for s_ in st])
        hs.set_stacks(stack("fx"), stack("fu"), stack("lx"), stack("lu"), stack("lxx"), stack("luu"), stack("lux"), stack("VxN"), stack("VxxN"))
        hs.set_constraint_stacks(stack("y"), stack("s"), stack("g"), stack("Gx"), stack("Gu"))
        reg0 = np.array([tw.reg for tw in twins]); mu = np.array([tw.mu for tw in twins])
        ok = hs.backward(api.STACKS_IPDDP_PATH, opt, reg0, mu, retry=True)
        K, k, Vx, Vxx, dV = hs.gains(); ky, Ky, ks, Ks, dX = hs.constraint_gains(); sc = hs.scalars()
        for b, tw in enumerate(twins):
            okt = False
            while not okt:          # the reference's retry loop (cddp_solver_base.cpp:93-111)
                okt = tw.backward()
                if not okt:
                    tw.reg_up()
                    if tw.reg_limit():
                        break
            assert bool(ok[b]) == okt and sc["reg"][b] == tw.reg, (kind, outer, b)
            if not okt:
                continue
            for name, got, ref in (("K", K[b], tw.K_u), ("k", k[b], tw.k_u), ("Vx", Vx[b], tw.Vx), ("Vxx", Vxx[b], tw.Vxx), ("dV", dV[b], tw.dV),
                                   ("k_y", ky[b], tw.k_y), ("K_y", Ky[b], tw.K_y), ("k_s", ks[b], tw.k_s), ("K_s", Ks[b], tw.K_s), ("dX", dX[b], tw.dX)):
                assert rel(got, ref) < TOL, (kind, outer, b, name, rel(got, ref))
            assert rel(sc["inf_du"][b], tw.inf_du) < TOL and rel(sc["inf_pr"][b], tw.inf_pr) < TOL
            assert rel(sc["inf_comp"][b], tw.inf_comp) < TOL and rel(sc["step_norm"][b], tw.step_norm) < TOL
            apm, adm = tw.max_step_sizes()
            assert abs(sc["alpha_pr_max"][b] - apm) < 1e-12 and abs(sc["alpha_du_max"][b] - adm) < 1e-12
            # the host side of the plugin loop: line search + apply with the twin (f(x, u) only exists on the host)
            best = tw.line_search()
            if best["success"]:
                tw.apply(best); tw.reg_down()
            else:
                tw.forward_failure()
    assert hs.kernel_ms() > 0.0
    hs.close()


@pytest.mark.parametrize("name", ["pendulum_ipddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box_ball", "cartpole_clddp_unc"])
def test_stack_fed_full_ddp_with_hessian_stacks(api, name):
    """options.use_ilqr = false for host plug-ins: the dt-scaled Hessian tensors F_xx, F_uu, F_ux (cddp_solver_base.cpp:346-356)
    go in through cddp_hip_set_hessian_stacks and every branch adds V_x(i) times them (ipddp_solver.cpp:1070-1082, 1396-1408);
    checked against the numpy twin's full-DDP sweep of the same iterate; dropping the stacks returns to Gauss-Newton."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_twin_golden as G
    B = 4
    rng = np.random.default_rng(20261130)
    twins, twins_gn = [], []
    for b in range(B):
        pert = None
        for ilqr, dst in ((False, twins), (True, twins_gn)):
            spec = G.CASES[name](); spec["options"]["use_ilqr"] = ilqr
            tw = G.T.Twin(spec)
            if pert is None: pert = (0.0 if b == 0 else 1.0) * rng.uniform(-0.02, 0.02, size=len(spec["x0"]))
            tw.set_initial(np.array(spec["x0"], float) + pert, spec.get("U0")); tw.initialize(); tw.X_lin, tw.U_lin = tw.X, tw.U
            dst.append(tw)
    tw0 = twins[0]
    hs = api.HipStackSolver(B, tw0.nx, tw0.nu, tw0.m, tw0.N)
    st = [_twin_stacks(tw) for tw in twins]
    stack = lambda key: np.stack([s_[key] for s_ in st])
    hs.set_stacks(stack("fx"), stack("fu"), stack("lx"), stack("lu"), stack("lxx"), stack("luu"), stack("lux"), stack("VxN"), stack("VxxN"))
    if tw0.m: hs.set_constraint_stacks(stack("y"), stack("s"), stack("g"), stack("Gx"), stack("Gu"))
    H = [[tw.hess_stack(t) for t in range(tw.N)] for tw in twins]
    Fxx = np.array([[h[0] for h in Hb] for Hb in H]); Fuu = np.array([[h[1] for h in Hb] for Hb in H]); Fux = np.array([[h[2] for h in Hb] for Hb in H])
    hs.set_hessian_stacks(Fxx, Fuu, Fux)
    opt = api.default_options()
    branch = api.STACKS_CLDDP if tw0.solver == "CLDDP" else (api.STACKS_IPDDP_PATH if tw0.m else api.STACKS_IPDDP)
    reg = np.array([tw.reg for tw in twins]); mu = np.array([tw.mu for tw in twins]) if tw0.m else None
    if branch == api.STACKS_CLDDP:      # CLDDPSolver::backwardPass has no second-order terms (clddp_solver.cpp:79-204): refused
        with pytest.raises(api.HipError, match="second-order"):
            hs.backward(branch, opt, reg, mu, retry=False)
    for pass_, ref in (("ddp", twins), ("gauss-newton", twins_gn)):
        if branch == api.STACKS_CLDDP and pass_ == "ddp": continue
        if pass_ == "gauss-newton": hs.set_hessian_stacks(None, None, None)
        ok = hs.backward(branch, opt, reg, mu, retry=False)
        K, k, Vx, Vxx, dV = hs.gains()
        for b, tw in enumerate(ref):
            okt = tw.backward()
            assert bool(ok[b]) == bool(okt), (name, pass_, b)
            if not okt: continue
            for nm, got, want in (("K", K[b], tw.K_u), ("k", k[b], tw.k_u), ("Vx", Vx[b], tw.Vx), ("Vxx", Vxx[b], tw.Vxx)):
                assert rel(got, want) < 1e-8, (name, pass_, b, nm, rel(got, want))      # (1e-8: indefinite Q_uu blocks under full DDP)
    # the tensor terms are not a no-op (trajectory 1 is off the equilibrium)
    if name.startswith("unicycle"): assert rel(twins[1].K_u, twins_gn[1].K_u) > 1e-8
    hs.close()


@pytest.mark.parametrize("name", ["pendulum_clddp_box", "cartpole_clddp_box", "unicycle_clddp_box"])
def test_stack_fed_clddp_with_control_box(api, name):
    """The control-limited step of CLDDP for host plug-ins (clddp_solver.cpp:147-178, boxqp.cpp:25-250): BoxQP per step on
    [lower - u_t, upper - u_t], warm-started with the previous sweep's k_t, feedback on the free directions -- against the
    numpy twin over the first iterations of its own solve (clamped and free steps both occur)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_twin_golden as G
    B = 4
    rng = np.random.default_rng(20261207)
    twins = []
    for b in range(B):
        spec = G.CASES[name]()
        tw = G.T.Twin(spec)
        x0 = np.array(spec["x0"], float) + (0.0 if b == 0 else 1.0) * rng.uniform(-0.05, 0.05, size=len(spec["x0"]))
        tw.set_initial(x0, spec.get("U0")); tw.initialize(); tw.X_lin, tw.U_lin = tw.X, tw.U
        twins.append(tw)
    tw0 = twins[0]
    box = tw0.clddp_box()
    assert box is not None
    hs = api.HipStackSolver(B, tw0.nx, tw0.nu, 0, tw0.N)
    opt = api.default_options()
    for key, val in tw0.o.items():
        if key.startswith("boxqp_") and hasattr(opt, key): setattr(opt, key, val)
    clamped = free = 0
    for outer in range(5):
        st = [_twin_stacks(tw) for tw in twins]
        stack = lambda key: np.stack([s_[key] for s_ in st])
        hs.set_stacks(stack("fx"), stack("fu"), stack("lx"), stack("lu"), stack("lxx"), stack("luu"), stack("lux"), stack("VxN"), stack("VxxN"))
        hs.set_control_box(box.lo if outer == 0 else None, box.up if outer == 0 else None, np.stack([tw.U for tw in twins]))
        reg0 = np.array([tw.reg for tw in twins])
        ok = hs.backward(api.STACKS_CLDDP, opt, reg0, None, retry=True)
        K, k, Vx, Vxx, dV = hs.gains(); sc = hs.scalars()
        for b, tw in enumerate(twins):
            okt = False
            while not okt:
                okt = tw.backward()
                if not okt:
                    tw.reg_up()
                    if tw.reg_limit(): break
            assert bool(ok[b]) == okt and sc["reg"][b] == tw.reg, (name, outer, b)
            if not okt: continue
            for nm, got, ref in (("K", K[b], tw.K_u), ("k", k[b], tw.k_u), ("Vx", Vx[b], tw.Vx), ("Vxx", Vxx[b], tw.Vxx), ("dV", dV[b], tw.dV)):
                assert rel(got, ref) < TOL, (name, outer, b, nm, rel(got, ref))
            assert rel(sc["inf_du"][b], tw.inf_du) < TOL
            rows = np.all(tw.K_u == 0.0, axis=2)          # a clamped control has a zero feedback row
            clamped += int(rows.sum()); free += int((~rows).sum())
            best = tw.line_search()
            if best["success"]: tw.apply(best); tw.reg_down()
            else: tw.forward_failure()
    assert clamped > 0 and free > 0, (clamped, free)
    # without the box the same handle is the unconstrained CLDDP sweep again
    hs.set_control_box(None, None, None)
    hs.backward(api.STACKS_CLDDP, opt, np.array([tw.reg for tw in twins]), None, retry=False)
    hs.close()


def test_stack_handle_argument_checks(api):
    hs = api.HipStackSolver(3, 1, 1, 2, 4)
    opt = api.default_options()
    with pytest.raises(api.HipError):
        hs.backward(api.STACKS_IPDDP_PATH, opt, 1e-6, 0.1)            # no stacks yet
    with pytest.raises(api.HipError):
        hs.set_stacks(fx=np.zeros((3, 4, 1, 1)))                       # first call must supply every stack
    with pytest.raises(api.HipError):
        api.HipStackSolver(3, 5, 5, 0, 4)                              # no instantiation
    z = lambda *sh: np.zeros(sh)
    hs.set_stacks(np.ones((3, 4, 1, 1)), np.ones((3, 4, 1, 1)), z(3, 4, 1), z(3, 4, 1), np.ones((3, 4, 1, 1)), np.ones((3, 4, 1, 1)), z(3, 4, 1, 1), z(3, 1), np.ones((3, 1, 1)))
    with pytest.raises(api.HipError):
        hs.backward(api.STACKS_IPDDP_PATH, opt, 1e-6, 0.1)            # constraint stacks missing
    with pytest.raises(api.HipError):
        hs.backward(api.STACKS_IPDDP, opt, 1e-6)                      # handle with m > 0 needs the path branch
    hs.set_constraint_stacks(np.ones((3, 4, 2)), np.ones((3, 4, 2)), -np.ones((3, 4, 2)), z(3, 4, 2, 1), np.ones((3, 4, 2, 1)))
    with pytest.raises(api.HipError):
        hs.backward(api.STACKS_IPDDP_PATH, opt, 1e-6, -1.0)           # non-positive mu
    ok = hs.backward(api.STACKS_IPDDP_PATH, opt, 1e-6, 0.1)
    assert ok.all()
    hs.close()
    # round-2 additions: control box, Hessian stacks, LogDDP branch
    h0 = api.HipStackSolver(3, 1, 1, 0, 4)
    h0.set_stacks(np.ones((3, 4, 1, 1)), np.ones((3, 4, 1, 1)), z(3, 4, 1), z(3, 4, 1), np.ones((3, 4, 1, 1)), np.ones((3, 4, 1, 1)), z(3, 4, 1, 1), z(3, 1), np.ones((3, 1, 1)))
    with pytest.raises(api.HipError):
        h0.set_control_box(np.array([-1.0]), None, z(3, 4, 1))         # bounds come together
    with pytest.raises(api.HipError):
        h0.set_control_box(None, None, z(3, 4, 1))                     # first call needs the bounds
    with pytest.raises(api.HipError):
        h0.set_control_box(np.array([1.0]), np.array([-1.0]), z(3, 4, 1))   # lower > upper
    with pytest.raises(api.HipError):
        h0.set_hessian_stacks(z(3, 4, 1, 1, 1), None, None)            # the three tensors come together
    h0.set_hessian_stacks(z(3, 4, 1, 1, 1), z(3, 4, 1, 1, 1), z(3, 4, 1, 1, 1))
    with pytest.raises(api.HipError, match="second-order"):
        h0.backward(api.STACKS_CLDDP, opt, 1e-6)                       # CLDDP has no tensor terms
    assert h0.backward(api.STACKS_LOGDDP, opt, 1e-6).all() and h0.backward(api.STACKS_IPDDP, opt, 1e-6).all()
    h0.set_hessian_stacks(None, None, None)
    h0.set_control_box(np.array([-1.0]), np.array([1.0]), z(3, 4, 1))
    assert h0.backward(api.STACKS_CLDDP, opt, 1e-6).all()
    with pytest.raises(api.HipError):
        h0.backward(7, opt, 1e-6)                                      # unknown branch
    h0.close()


# ---- round 6: the terminal-equality branch of the stack-fed sweeps (stacks_te.hpp, CDDP_HIP_STACKS_IPDDP_TERM_EQ) -------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 1, 2), (1, 1, 1, 8), (2, 1, 2, 12), (3, 2, 2, 20), (4, 1, 3, 15)])
def test_terminal_equality_stack_sweep_against_the_twin(api, shape):
    """solveTerminalEqualityLQR (ipddp_solver.cpp:478-639) on host-fed LQ stacks: dense H_T, previous multipliers, cross terms M, indefinite-free
    random models -- gains, value recursion (P, p), multiplier step, linear-policy rollout, max |r + B^T p| and max |k| against the numpy twin's
    restatement (oracle/twin/cddp_twin_te.py), per trajectory."""
    import cddp_twin_te as TE
    nx, nu, pT, N = shape
    B = 5
    rng = np.random.default_rng(nx * 100 + nu * 10 + pT)
    A = np.tile(np.eye(nx), (B, N, 1, 1)) + 0.1 * rng.standard_normal((B, N, nx, nx)); Bm = 0.3 * rng.standard_normal((B, N, nx, nu))
    def spd(n, lead):
        W = rng.standard_normal(lead + (n, n)); return np.einsum("...ij,...kj->...ik", W, W) / n + 0.5 * np.eye(n)
    Q = spd(nx, (B, N)); R = spd(nu, (B, N)); QN = spd(nx, (B,))
    q = rng.standard_normal((B, N, nx)); r = rng.standard_normal((B, N, nu)); qN = rng.standard_normal((B, nx))
    M = 0.05 * rng.standard_normal((B, N, nx, nu))
    HT = rng.standard_normal((B, pT, nx)); bT = rng.standard_normal((B, pT)); lam_prev = 0.3 * rng.standard_normal((B, pT))
    mu, rs, rexp, reg = 0.05, 1e-8, 0.25, 1e-6
    floor = max(1e-10, rs * mu ** rexp)
    hs = api.HipStackSolver(B, nx, nu, 0, N)
    hs.set_stacks(A, Bm, q, r, Q, R, M, qN, QN)                    # fx = A, fu = B, lx = q, lu = r, lxx = Q, luu = R, lux = M (nx x nu), VxN = q_N, VxxN = Q_N
    hs.set_terminal_equality(HT, bT, lam_prev, floor)
    ok = hs.backward(api.STACKS_IPDDP_TERM_EQ, api.default_options(), reg, None, retry=False)
    K, k, p, P, dV = hs.gains()
    sc = hs.scalars()
    dlam, dX = hs.terminal()
    hs.close()
    assert ok.all()
    for b in range(B):
        Rr = [R[b, t] + reg * np.eye(nu) for t in range(N)]
        okt, Kt, kt, Pt, pt, lam_tot, lam_d = TE.terminal_equality_lqr([Q[b, t] for t in range(N)] + [QN[b]], [q[b, t] for t in range(N)] + [qN[b]], Rr, [r[b, t] for t in range(N)],
                                                                       [M[b, t] for t in range(N)], [A[b, t] for t in range(N)], [Bm[b, t] for t in range(N)], [np.zeros(nx)] * N,
                                                                       np.zeros(nx), HT[b], bT[b], mu, rs, rexp, lam_prev[b])
        assert okt
        tol = lambda ref: 1e-9 * max(1.0, float(np.max(np.abs(ref))))
        assert np.max(np.abs(K[b] - np.stack(Kt))) < tol(np.stack(Kt)) and np.max(np.abs(k[b] - np.stack(kt))) < tol(np.stack(kt))
        assert np.max(np.abs(P[b] - np.stack(Pt))) < tol(np.stack(Pt)) and np.max(np.abs(p[b] - np.stack(pt))) < tol(np.stack(pt))
        assert np.max(np.abs(dlam[b] - lam_d)) < tol(lam_d)
        dXt, _ = TE.rollout_linear([A[b, t] for t in range(N)], [Bm[b, t] for t in range(N)], [np.zeros(nx)] * N, Kt, kt, np.zeros(nx))
        assert np.max(np.abs(dX[b] - np.stack(dXt))) < tol(np.stack(dXt))
        inf_du = max(float(np.max(np.abs(r[b, t] + Bm[b, t].T @ pt[t + 1]))) for t in range(N))
        assert abs(sc["inf_du"][b] - inf_du) < 1e-9 * max(1.0, inf_du) and abs(sc["step_norm"][b] - max(float(np.max(np.abs(v))) for v in kt)) < 1e-9 * max(1.0, float(np.max(np.abs(np.stack(kt)))))
