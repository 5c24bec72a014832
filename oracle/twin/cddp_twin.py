"""numpy twin of the reference's CLDDP / IPDDP path -- TEST INFRASTRUCTURE.

A second, independently written restatement of astomodynamics/cddp-cpp's solver core (the first is the C++ oracle
in oracle/cddp_oracle.cpp).  It was written from the reference sources alone -- each function cites the reference
lines it follows -- WITHOUT consulting oracle/ or the HIP kernels, so that a misreading of the reference would have
to be made twice, independently, to go unnoticed.  tests/golden/make_twin_golden.py runs it (in the build container
only) and commits its vectors; tests/test_twin_golden.py holds the C++ oracle and the HIP path against them.

It does NOT lift "parity unpinned": Eigen 3.4 / autodiff 1.1.2 are not available here, so the Eigen pieces that
steer solver decisions (LDLT with diagonal pivoting, its solve with the D^+ threshold) are restated below from the
Eigen 3.4 algorithm; the eigenvalue test and the dense inverse of CLDDP use LAPACK through numpy.

Covered: CLDDP (+BoxQP), IPDDP unconstrained / path-constrained / terminal inequality / terminal equality, the
forward pass + filter, barrier update, convergence tests, the outer loop; pendulum, cart-pole, unicycle, LTI plants;
control box, state box, ball, linear path constraints.  Python loops: small problems only.

All citations are relative to /root/reference/.
"""
import math
import sys

import numpy as np

INF = float("inf")


def _sin(v):          # math.sin raises on a non-finite argument; a diverged rollout must yield NaN like std::sin
    return math.sin(v) if math.isfinite(v) else float("nan")


def _cos(v):
    return math.cos(v) if math.isfinite(v) else float("nan")

# ipddp_solver.cpp:33-38
K_SLACK_INTERIOR_OFFSET = 1e-4
EPS_SLACK = 1e-10
EPS_DUAL = 1e-10
MAX_BARRIER_RATIO = 1e6

STATUS = ["Running", "OptimalSolutionFound", "AcceptableSolutionFound", "MaxIterationsReached",
          "RegularizationLimitReached_NotConverged", "MaxCpuTimeReached"]


def default_options():
    """include/cddp-cpp/cddp_core/options.hpp:41-251 and boxqp.hpp:30-41 (in-class initialisers)."""
    return dict(
        tolerance=1e-5, acceptable_tolerance=1e-6, max_iterations=1, use_ilqr=True, enable_parallel=False,
        termination_scaling_max_factor=100.0,
        ls_max_iterations=11, ls_initial_step_size=1.0, ls_min_step_size=1e-8, ls_step_reduction_factor=0.5,
        reg_initial_value=1e-6, reg_update_factor=10.0, reg_max_value=1e7, reg_min_value=1e-10,
        boxqp_max_iterations=100, boxqp_min_gradient_norm=1e-8, boxqp_min_relative_improvement=1e-8,
        boxqp_step_decrease_factor=0.6, boxqp_min_step_size=1e-22, boxqp_armijo_constant=0.1,
        filter_merit_acceptance_threshold=1e-6, filter_violation_acceptance_threshold=1e-6,
        filter_max_violation_threshold=1e4, filter_min_violation_for_armijo_check=1e-7, filter_armijo_constant=1e-4,
        dual_var_init_scale=0.1, slack_var_init_scale=1e-2, barrier_tol_mult=0.1, barrier_update_dual_weight=0.01,
        mu_kappa_epsilon=10.0, check_state_stationarity=False, theta_norm="l1", max_filter_size=5,
        theta_0_floor=1.0, jacobian_regularization_value=1e-8, jacobian_regularization_exponent=0.25,
        mu_initial=1.0, mu_min_value=1e-10, mu_update_factor=0.5, mu_update_power=1.2,
        min_fraction_to_boundary=0.99, barrier_strategy="ADAPTIVE")


# --------------------------------------------------------------------------------------------------------------
# Eigen pieces
# --------------------------------------------------------------------------------------------------------------
class EigenLDLT:
    """Eigen 3.4 LDLT<MatrixXd, Lower> (Eigen/src/Cholesky/LDLT.h: ldlt_inplace<Lower>::unblocked and
    LDLT::_solve_impl), restated: symmetric pivoting on the largest |diagonal| entry, no square roots,
    info() == Success unless a non-zero pivot follows a zero pivot (or non-zero entries sit under a zero pivot) --
    an INDEFINITE matrix is accepted; solve() zeroes the components whose |D_ii| <= DBL_MIN."""

    def __init__(self, A):
        A = np.array(A, dtype=np.float64)
        n = A.shape[0]
        M = A.copy()              # only the lower triangle is referenced / updated
        tr = np.arange(n)
        ok = True
        found_zero_pivot = False
        for k in range(n):
            d = np.abs(np.diag(M)[k:])
            p = k + int(np.argmax(d))          # first maximum, as maxCoeff(&index)
            tr[k] = p
            if p != k:
                s = n - p - 1
                M[[k, p], :k] = M[[p, k], :k]                     # row(k).head(k) <-> row(p).head(k)
                if s > 0:
                    tmp = M[p + 1:, k].copy(); M[p + 1:, k] = M[p + 1:, p]; M[p + 1:, p] = tmp
                M[k, k], M[p, p] = M[p, p], M[k, k]
                for i in range(k + 1, p):                          # the strip between the two pivots
                    tmp = M[i, k]; M[i, k] = M[p, i]; M[p, i] = tmp
            rs = n - k - 1
            if k > 0:
                temp = np.diag(M)[:k] * M[k, :k]
                M[k, k] -= float(M[k, :k] @ temp)
                if rs > 0:
                    M[k + 1:, k] -= M[k + 1:, :k] @ temp
            akk = M[k, k]
            pivot_is_valid = abs(akk) > 0.0
            if k == 0 and not pivot_is_valid:                      # whole diagonal zero
                tr = np.arange(n)
                for j in range(n):
                    ok = ok and bool(np.all(M[j + 1:, j] == 0.0))
                break
            if rs > 0 and pivot_is_valid:
                M[k + 1:, k] /= akk
            elif rs > 0:
                ok = ok and bool(np.all(M[k + 1:, k] == 0.0))
            if found_zero_pivot and pivot_is_valid:
                ok = False
            elif not pivot_is_valid:
                found_zero_pivot = True
        self.n, self.M, self.tr, self.ok = n, M, tr, ok

    def solve(self, B):
        B = np.array(B, dtype=np.float64)
        vec = B.ndim == 1
        X = B.reshape(self.n, -1).copy()
        n, M = self.n, self.M
        for k in range(n):                                         # dst = P b
            p = self.tr[k]
            if p != k:
                X[[k, p], :] = X[[p, k], :]
        for i in range(n):                                         # L^-1 (unit lower)
            for j in range(i):
                X[i, :] -= M[i, j] * X[j, :]
        tol = np.finfo(np.float64).tiny
        for i in range(n):                                         # pseudo-inverse of D
            if abs(M[i, i]) > tol:
                X[i, :] /= M[i, i]
            else:
                X[i, :] = 0.0
        for i in range(n - 1, -1, -1):                             # L^-T
            for j in range(i + 1, n):
                X[i, :] -= M[j, i] * X[j, :]
        for k in range(n - 1, -1, -1):                             # P^T
            p = self.tr[k]
            if p != k:
                X[[k, p], :] = X[[p, k], :]
        return X[:, 0] if vec else X.reshape(B.shape)


def sym(M):                      # symmetrizeMatrix, ipddp_solver.cpp:217-220
    return 0.5 * (M + M.T)


def clip_pos(num, den):          # clipPositiveBarrierRatio :222-225
    return min(max(num / den, 0.0), MAX_BARRIER_RATIO)


def clip_sgn(num, den):          # clipSignedBarrierRatio :227-231
    return min(max(num / den, -MAX_BARRIER_RATIO), MAX_BARRIER_RATIO)


# --------------------------------------------------------------------------------------------------------------
# Plants: continuous dynamics + continuous Jacobians, each with the reference's own derivative source
# --------------------------------------------------------------------------------------------------------------
class Pendulum:      # src/dynamics_model/pendulum.cpp:29-66 (double path: +sin, theta = 0 upright)
    nx, nu = 2, 1

    def __init__(self, length, mass, damping, gravity=9.81):
        self.l, self.m, self.b, self.g = length, mass, damping, gravity

    def f(self, x, u, t):
        inertia = self.m * self.l * self.l
        return np.array([x[1], (u[0] - self.b * x[1] + self.m * self.g * self.l * _sin(x[0])) / inertia])

    def jac(self, x, u, t):
        A = np.zeros((2, 2)); B = np.zeros((2, 1))
        A[0, 1] = 1.0
        A[1, 0] = (self.g / self.l) * _cos(x[0])
        A[1, 1] = -self.b / (self.m * self.l * self.l)
        B[1, 0] = 1.0 / (self.m * self.l * self.l)
        return A, B

    def hess(self, x, u, t):
        """getStateHessian / getControlHessian are analytic (pendulum.cpp:68-85); getCrossHessian is NOT overridden, so it is
        autodiff of getContinuousDynamicsAutodiff (the -sin variant, :87-100) -- whose u-x cross derivatives are zero."""
        Fxx, Fuu, Fux = _zeros_hess(2, 1)
        Fxx[1, 0, 0] = -(self.g / self.l) * _sin(x[0])
        return Fxx, Fuu, Fux


def _zeros_hess(nx, nu):
    return np.zeros((nx, nx, nx)), np.zeros((nx, nu, nu)), np.zeros((nx, nu, nx))


class CartPole:      # src/dynamics_model/cartpole.cpp:38-103; Jacobians = exact derivatives of the AUTODIFF twin (:69-93,
    nx, nu = 4, 1    # which carries the damping term the double path omits)

    def __init__(self, mc, mp, l, g, damping):
        self.mc, self.mp, self.l, self.g, self.d = mc, mp, l, g, damping
        self._jac = None

    def f(self, x, u, t):
        th, xd, thd, F = x[1], x[2], x[3], u[0]
        s, c = _sin(th), _cos(th)
        den = self.mc + self.mp * s * s
        return np.array([xd, thd,
                         (F + self.mp * s * (self.l * thd * thd + self.g * c)) / den,
                         (-F * c - self.mp * self.l * thd * thd * c * s - (self.mc + self.mp) * self.g * s) / (self.l * den)])

    def _build(self):
        import sympy as sp
        X, TH, XD, THD, F = sp.symbols("x th xd thd F")
        s, c = sp.sin(TH), sp.cos(TH)
        den = self.mc + self.mp * s * s
        fx = [XD, THD, (F + self.mp * s * (self.l * THD * THD + self.g * c)) / den,
              (-F * c - self.mp * self.l * THD * THD * c * s - (self.mc + self.mp) * self.g * s - self.d * THD) / (self.l * den)]
        J = sp.Matrix(fx).jacobian([X, TH, XD, THD, F])
        self._jac = sp.lambdify([X, TH, XD, THD, F], [[J[i, j] for j in range(5)] for i in range(4)], "math")

    def jac(self, x, u, t):
        if self._jac is None:
            self._build()
        J = np.array(self._jac(x[0], x[1], x[2], x[3], u[0]), dtype=np.float64)
        return J[:, :4].copy(), J[:, 4:].copy()

    def hess(self, x, u, t):
        """DynamicalSystem::getStateHessian / getControlHessian / getCrossHessian (dynamical_system.cpp:137-217): second
        derivatives of the autodiff path w.r.t. z = [x, u]; blocks H[:n, :n], H[n:, n:], H[n:, :n]."""
        if getattr(self, "_hess", None) is None:
            import sympy as sp
            X, TH, XD, THD, F = sp.symbols("x th xd thd F")
            sn, cs = sp.sin(TH), sp.cos(TH)
            den = self.mc + self.mp * sn * sn
            fx = [XD, THD, (F + self.mp * sn * (self.l * THD * THD + self.g * cs)) / den,
                  (-F * cs - self.mp * self.l * THD * THD * cs * sn - (self.mc + self.mp) * self.g * sn - self.d * THD) / (self.l * den)]
            z = [X, TH, XD, THD, F]
            H = [[[sp.diff(fi, a, b) for b in z] for a in z] for fi in fx]
            self._hess = sp.lambdify(z, H, "math")
        H = np.array(self._hess(x[0], x[1], x[2], x[3], u[0]), dtype=np.float64)     # (4, 5, 5)
        return H[:, :4, :4].copy(), H[:, 4:, 4:].copy(), H[:, 4:, :4].copy()


class Unicycle:      # src/dynamics_model/unicycle.cpp:28-66
    nx, nu = 3, 2

    def f(self, x, u, t):
        return np.array([u[0] * _cos(x[2]), u[0] * _sin(x[2]), u[1]])

    def jac(self, x, u, t):
        A = np.zeros((3, 3)); B = np.zeros((3, 2))
        A[0, 2] = -u[0] * _sin(x[2]); A[1, 2] = u[0] * _cos(x[2])
        B[0, 0] = _cos(x[2]); B[1, 0] = _sin(x[2]); B[2, 1] = 1.0
        return A, B

    def hess(self, x, u, t):
        """State Hessian analytic (unicycle.cpp:68-80), control Hessian zero (:82-89); the cross Hessian is the autodiff default
        on getContinuousDynamicsAutodiff (:91-107): d2(v cos th)/dv dth = -sin th, d2(v sin th)/dv dth = cos th."""
        Fxx, Fuu, Fux = _zeros_hess(3, 2)
        Fxx[0, 2, 2] = -u[0] * _cos(x[2]); Fxx[1, 2, 2] = -u[0] * _sin(x[2])
        Fux[0, 0, 2] = -_sin(x[2]); Fux[1, 0, 2] = _cos(x[2])
        return Fxx, Fuu, Fux


class Bicycle:       # src/dynamics_model/bicycle.cpp:28-156: state [x, y, theta, v], control [a, delta]
    nx, nu = 4, 2

    def __init__(self, wheelbase):
        self.L = wheelbase

    def f(self, x, u, t):
        return np.array([x[3] * _cos(x[2]), x[3] * _sin(x[2]), (x[3] / self.L) * math.tan(u[1]), u[0]])

    def jac(self, x, u, t):      # :68-111 analytic
        A = np.zeros((4, 4)); B = np.zeros((4, 2))
        th, v, dl = x[2], x[3], u[1]
        A[0, 2] = -v * _sin(th); A[0, 3] = _cos(th); A[1, 2] = v * _cos(th); A[1, 3] = _sin(th); A[2, 3] = math.tan(dl) / self.L
        B[3, 0] = 1.0; B[2, 1] = v / (self.L * _cos(dl) ** 2)
        return A, B

    def hess(self, x, u, t):     # state / control Hessians :113-156 analytic; cross Hessian: the base-class autodiff default, in closed form
        Fxx, Fuu, Fux = _zeros_hess(4, 2)
        th, v, dl = x[2], x[3], u[1]
        Fxx[0, 2, 2] = -v * _cos(th); Fxx[0, 2, 3] = Fxx[0, 3, 2] = -_sin(th)
        Fxx[1, 2, 2] = -v * _sin(th); Fxx[1, 2, 3] = Fxx[1, 3, 2] = _cos(th)
        Fuu[2, 1, 1] = 2.0 * v * _sin(dl) / (self.L * _cos(dl) ** 3)
        Fux[2, 1, 3] = 1.0 / (self.L * _cos(dl) ** 2)          # d2 (v tan(delta) / L) / d delta d v
        return Fxx, Fuu, Fux


class HCW:           # src/dynamics_model/spacecraft_linear.cpp:24-120: Hill-Clohessy-Wiltshire relative motion, linear time-invariant
    nx, nu = 6, 3

    def __init__(self, mean_motion, mass):
        n = mean_motion
        self.A = np.zeros((6, 6)); self.B = np.zeros((6, 3))
        self.A[0, 3] = self.A[1, 4] = self.A[2, 5] = 1.0
        n2 = n * n                                           # the reference forms n2 first (spacecraft_linear.cpp:49, 60)
        self.A[3, 0] = 3.0 * n2; self.A[3, 4] = 2.0 * n; self.A[4, 3] = -2.0 * n; self.A[5, 2] = -n2
        self.B[3, 0] = self.B[4, 1] = self.B[5, 2] = 1.0 / mass
        self.n, self.mass = n, mass

    def f(self, x, u, t):
        n = self.n; n2 = n * n
        return np.array([x[3], x[4], x[5], 2.0 * n * x[4] + 3.0 * n2 * x[0] + u[0] / self.mass, -2.0 * n * x[3] + u[1] / self.mass,
                         -n2 * x[2] + u[2] / self.mass])

    def jac(self, x, u, t):
        return self.A.copy(), self.B.copy()

    def hess(self, x, u, t):
        return _zeros_hess(6, 3)


class Car:           # src/dynamics_model/car.cpp: a DISCRETE plant; state [x, y, theta, v], control [delta, a]
    nx, nu = 4, 2
    discrete = True

    def __init__(self, wheelbase, dt):
        self.d, self.h = wheelbase, dt

    def step(self, x, u, t):     # :24-60
        d, h = self.d, self.h
        f = h * x[3]
        b = d + f * _cos(u[0]) - math.sqrt(d * d - (f * _sin(u[0])) ** 2)
        return x + np.array([b * _cos(x[2]), b * _sin(x[2]), math.asin(_sin(u[0]) * f / d), h * u[1]])

    def _parts(self, x, u):
        """First and second partials of b(f, delta) and phi(v, delta) -- the hand-derived derivatives of what the reference
        differentiates with autodiff (:62-161), including its two clamps (:183-186, 196-199)."""
        d, h = self.d, self.h
        f = h * x[3]; s, c = _sin(u[0]), _cos(u[0])
        I = d * d - (f * s) ** 2
        clamped = I < 0.0
        if clamped:
            I = 0.0
        r = math.sqrt(I)
        if clamped or r == 0.0:
            rf = rd = rff = rfd = rdd = 0.0 if clamped else float("inf")
        else:
            If, Id = -2.0 * f * s * s, -2.0 * f * f * s * c
            Iff, Ifd, Idd = -2.0 * s * s, -4.0 * f * s * c, -2.0 * f * f * (c * c - s * s)
            rf, rd = If / (2 * r), Id / (2 * r)
            rff = Iff / (2 * r) - If * If / (4 * r ** 3); rfd = Ifd / (2 * r) - If * Id / (4 * r ** 3); rdd = Idd / (2 * r) - Id * Id / (4 * r ** 3)
        b = d + f * c - r
        bv, bd = h * (c - rf), -f * s - rd
        bvv, bvd, bdd = h * h * (-rff), h * (-s - rfd), -f * c - rdd
        w = s * f / d
        if abs(w) > 1.0:      # clamped argument: a constant
            pv = pd = pvv = pvd = pdd = 0.0
        else:
            p1 = 1.0 / math.sqrt(1.0 - w * w); p2 = w / (1.0 - w * w) ** 1.5
            wv, wd, wvd, wdd = s * h / d, c * f / d, c * h / d, -w
            pv, pd = p1 * wv, p1 * wd
            pvv, pvd, pdd = p2 * wv * wv, p2 * wv * wd + p1 * wvd, p2 * wd * wd + p1 * wdd
        return b, bv, bd, bvv, bvd, bdd, pv, pd, pvv, pvd, pdd

    def jac(self, x, u, t):      # :62-111: gradient of the discrete map, J.diagonal() -= 1, J /= timestep
        b, bv, bd, _, _, _, pv, pd, _, _, _ = self._parts(x, u)
        ct, st = _cos(x[2]), _sin(x[2])
        A = np.eye(4); B = np.zeros((4, 2))
        A[0, 2] = -b * st; A[0, 3] = ct * bv; A[1, 2] = b * ct; A[1, 3] = st * bv; A[2, 3] = pv
        B[0, 0] = ct * bd; B[1, 0] = st * bd; B[2, 0] = pd; B[3, 1] = self.h
        return (A - np.eye(4)) / self.h, B / self.h

    def hess(self, x, u, t):     # :113-161 (+ the base-class cross Hessian on (x+ - x) / timestep): Hessian of the discrete map / timestep
        b, bv, bd, bvv, bvd, bdd, pv, pd, pvv, pvd, pdd = self._parts(x, u)
        ct, st = _cos(x[2]), _sin(x[2])
        Fxx, Fuu, Fux = _zeros_hess(4, 2)
        Fxx[0, 2, 2] = -b * ct; Fxx[0, 2, 3] = Fxx[0, 3, 2] = -st * bv; Fxx[0, 3, 3] = ct * bvv
        Fxx[1, 2, 2] = -b * st; Fxx[1, 2, 3] = Fxx[1, 3, 2] = ct * bv; Fxx[1, 3, 3] = st * bvv
        Fxx[2, 3, 3] = pvv
        Fuu[0, 0, 0] = ct * bdd; Fuu[1, 0, 0] = st * bdd; Fuu[2, 0, 0] = pdd
        Fux[0, 0, 2] = -st * bd; Fux[0, 0, 3] = ct * bvd
        Fux[1, 0, 2] = ct * bd; Fux[1, 0, 3] = st * bvd
        Fux[2, 0, 3] = pvd
        return Fxx / self.h, Fuu / self.h, Fux / self.h


class LTI:           # src/dynamics_model/lti_system.cpp:71-92: discrete x+ = A x + B u; Jacobians (A - I)/dt, B/dt
    def __init__(self, A, B, dt):
        self.A, self.B, self.dt = np.array(A, float), np.array(B, float), dt
        self.nx, self.nu = self.A.shape[0], self.B.shape[1]
        self.discrete = True

    def step(self, x, u, t):
        return self.A @ x + self.B @ u

    def jac(self, x, u, t):
        return (self.A - np.eye(self.nx)) / self.dt, self.B / self.dt

    def hess(self, x, u, t):           # lti_system.cpp:94-115: zero
        return _zeros_hess(self.nx, self.nu)


def discrete_step(model, integrator, dt, x, u, t):
    """DynamicalSystem::getDiscreteDynamics (src/cddp_core/dynamical_system.cpp:28-83)."""
    if getattr(model, "discrete", False):
        return model.step(x, u, t)
    f = model.f
    if integrator == "euler":
        return x + dt * f(x, u, t)
    if integrator == "heun":
        k1 = f(x, u, t); k2 = f(x + dt * k1, u, t + dt)
        return x + 0.5 * dt * (k1 + k2)
    if integrator == "rk3":
        k1 = f(x, u, t); k2 = f(x + 0.5 * dt * k1, u, t + 0.5 * dt); k3 = f(x - dt * k1 + 2 * dt * k2, u, t + dt)
        return x + (dt / 6) * (k1 + 4 * k2 + k3)
    if integrator == "rk4":
        k1 = f(x, u, t); k2 = f(x + 0.5 * dt * k1, u, t + 0.5 * dt); k3 = f(x + 0.5 * dt * k2, u, t + 0.5 * dt)
        k4 = f(x + dt * k3, u, t + dt)
        return x + (dt / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
    raise ValueError(integrator)


# --------------------------------------------------------------------------------------------------------------
# Constraints: evaluate(x, u) - getUpperBound(), state / control Jacobians (include/cddp-cpp/cddp_core/constraint.hpp)
# --------------------------------------------------------------------------------------------------------------
class ControlBox:    # BoxConstraint<Control> :144-251
    def __init__(self, lower, upper, scale=1.0):
        self.lo, self.up, self.scale = np.array(lower, float), np.array(upper, float), scale
        self.n = self.up.size; self.dim = 2 * self.n
        self.ip_upper = np.concatenate([-self.lo * scale, self.up * scale])

    def g(self, x, u):
        return np.concatenate([-u, u]) * self.scale - self.ip_upper

    def jac(self, x, u):
        Gx = np.zeros((self.dim, x.size)); Gu = np.zeros((self.dim, u.size))
        Gu[:self.n, :] = -np.eye(self.n) * self.scale; Gu[self.n:, :] = np.eye(self.n) * self.scale
        return Gx, Gu


class StateBox(ControlBox):   # BoxConstraint<State>
    def g(self, x, u):
        return np.concatenate([-x, x]) * self.scale - self.ip_upper

    def jac(self, x, u):
        Gx = np.zeros((self.dim, x.size)); Gu = np.zeros((self.dim, u.size))
        Gx[:self.n, :] = -np.eye(self.n) * self.scale; Gx[self.n:, :] = np.eye(self.n) * self.scale
        return Gx, Gu


class Ball:          # BallConstraint :313-404: g = -scale |x[:d] - c|^2, upper = -scale r^2
    dim = 1

    def __init__(self, radius, center, scale=1.0):
        self.r, self.c, self.scale = radius, np.array(center, float), scale

    def g(self, x, u):
        diff = x[:self.c.size] - self.c
        return np.array([-(self.scale * float(diff @ diff))]) - np.array([-(self.r * self.r) * self.scale])

    def jac(self, x, u):
        Gx = np.zeros((1, x.size)); Gu = np.zeros((1, u.size))
        diff = x[:self.c.size] - self.c
        Gx[0, :self.c.size] = -2.0 * self.scale * diff
        return Gx, Gu


class Linear:        # LinearConstraint :253-311: g = A x, upper = b
    def __init__(self, A, b):
        self.A, self.b = np.array(A, float), np.array(b, float); self.dim = self.b.size

    def g(self, x, u):
        return self.A @ x - self.b

    def jac(self, x, u):
        return self.A.copy(), np.zeros((self.dim, u.size))


class SecondOrderCone:   # SecondOrderConeConstraint :626-800: g = cos(fov) sqrt(|p - o|^2 + eps) - (p - o) . axis, p = x[:3]; upper 0
    dim = 1

    def __init__(self, origin, direction, fov, eps=1e-6):
        if fov < 0 or fov > math.pi:
            raise ValueError("SecondOrderConeConstraint: Cone angle must be between 0 and PI.")
        if eps <= 0:
            raise ValueError("SecondOrderConeConstraint: Regularization epsilon must be positive.")
        a = np.array(direction, float); n = math.sqrt(float(a @ a))
        if n == 0.0:
            raise ValueError("SecondOrderConeConstraint: Opening direction cannot be zero vector.")
        self.o, self.ax, self.cosf, self.eps = np.array(origin, float), a / n, math.cos(fov), eps

    def g(self, x, u):
        v = x[:3] - self.o
        return np.array([math.sqrt(float(v @ v) + self.eps) * self.cosf - float(v @ self.ax)])

    def jac(self, x, u):
        Gx = np.zeros((1, x.size)); Gu = np.zeros((1, u.size))
        v = x[:3] - self.o; rn = math.sqrt(float(v @ v) + self.eps)
        Gx[0, :3] = self.cosf * (v / rn) - self.ax if rn > 1e-9 else -self.ax
        return Gx, Gu


class ThrustMagnitude:   # ThrustMagnitudeConstraint :802-927 (two rows) / MaxThrustMagnitudeConstraint :929-1048 (min_norm None: one row)
    def __init__(self, min_norm, max_norm, eps=1e-6):
        self.mn, self.mx, self.eps = min_norm, max_norm, eps
        self.dim = 1 if min_norm is None else 2

    def g(self, x, u):
        n = math.sqrt(float(u @ u))
        return np.array([n - self.mx]) if self.mn is None else np.array([self.mn - n, n - self.mx])

    def jac(self, x, u):     # the Jacobian uses the REGULARISED norm, the value the plain one
        Gx = np.zeros((self.dim, x.size)); Gu = np.zeros((self.dim, u.size))
        rn = math.sqrt(float(u @ u) + self.eps)
        if self.mn is None:
            if rn > sys.float_info.min:
                Gu[0] = u / rn
        elif not rn < self.eps:
            Gu[0] = -(u / rn); Gu[1] = u / rn
        return Gx, Gu


# --------------------------------------------------------------------------------------------------------------
# BoxQP (src/cddp_core/boxqp.cpp:25-250)
# --------------------------------------------------------------------------------------------------------------
def boxqp(H, g, lower, upper, x0, o):
    n = g.size
    x = np.minimum(np.maximum(x0, lower), upper) if x0 is not None and x0.size == n else 0.5 * (lower + upper)
    clamped = np.zeros(n, dtype=bool)
    free = np.ones(n, dtype=bool)
    obj = lambda z: 0.5 * float(z @ (H @ z)) + float(g @ z)
    value = obj(x); old_value = INF
    status = "MAX_ITER_EXCEEDED"; fac = None
    for it in range(o["boxqp_max_iterations"]):
        if it > 0 and abs(old_value - value) < o["boxqp_min_relative_improvement"] * abs(old_value):
            status = "SUCCESS"; break
        old_value = value
        grad = g + H @ x
        old_clamped = clamped
        clamped = ((x == lower) & (grad > 0)) | ((x == upper) & (grad < 0))
        free = ~clamped
        if clamped.all():
            status = "ALL_CLAMPED"; break
        if it == 0 or np.any(old_clamped != clamped):
            idx = np.where(free)[0]
            fac = EigenLDLT(H[np.ix_(idx, idx)])
            if not fac.ok:
                status = "HESSIAN_NOT_PD"; break
        gnorm = math.sqrt(float(np.sum(grad[free] ** 2)))
        if gnorm < o["boxqp_min_gradient_norm"]:
            status = "SUCCESS"; break
        grad_clamped = g.copy()
        for i in range(n):
            if clamped[i]:
                grad_clamped = grad_clamped + H[:, i] * x[i]
        idx = np.where(free)[0]
        search = np.zeros(n)
        search[idx] = -fac.solve(grad_clamped[idx]) - x[idx]
        sdotg = float(search @ grad)
        if sdotg >= 0:
            status = "NO_DESCENT"; break
        step = 1.0; found = False
        while step > o["boxqp_min_step_size"]:
            xn = np.minimum(np.maximum(x + step * search, lower), upper)
            vn = obj(xn)
            if (vn - value) <= o["boxqp_armijo_constant"] * step * sdotg:
                found = True; break
            step *= o["boxqp_step_decrease_factor"]
        if not found:
            status = "MAX_LS_EXCEEDED"; break
        x = xn; value = obj(x)
    return x, status, free, fac


# --------------------------------------------------------------------------------------------------------------
# Solver
# --------------------------------------------------------------------------------------------------------------
class Twin:
    """cddp::CDDP + CLDDPSolver / IPDDPSolver for one trajectory.

    spec keys: solver ("CLDDP"|"IPDDP"), model, integrator, dt, N, Q, R, Qf, xref, [xref_traj],
    constraints {name: obj} (iterated in sorted-name order = std::map order), terminal {name: ("eq", target) |
    ("ineq", A, b)}, options (dict overriding default_options())."""

    def __init__(self, spec):
        self.solver = spec["solver"]; self.model = spec["model"]; self.integrator = spec["integrator"]
        self.dt = spec["dt"]; self.N = spec["N"]
        self.nx, self.nu = self.model.nx, self.model.nu
        self.Qdt = np.array(spec["Q"], float) * self.dt            # objective.cpp:38-39
        self.Rdt = np.array(spec["R"], float) * self.dt
        self.Qf = np.array(spec["Qf"], float)
        self.xref = np.array(spec["xref"], float)
        self.xref_traj = spec.get("xref_traj")
        self.o = default_options(); self.o.update(spec.get("options", {}))
        cons = spec.get("constraints", {})
        self.cons = [(k, cons[k]) for k in sorted(cons)]            # std::map<std::string, ...> order
        self.m = sum(c.dim for _, c in self.cons)
        term = spec.get("terminal", {})
        self.term_ineq = [(k, np.array(term[k][1], float), np.array(term[k][2], float)) for k in sorted(term) if term[k][0] == "ineq"]
        self.term_eq = [(k, np.array(term[k][1], float)) for k in sorted(term) if term[k][0] == "eq"]
        self.pT = sum(t.size for _, t in self.term_eq)
        self.alphas = self.build_alphas()
        self.history = []

    # ---- detail::buildLineSearchAlphas (cddp_context_utils.cpp:37-57)
    def build_alphas(self):
        o = self.o; out = []; cur = o["ls_initial_step_size"]
        for i in range(o["ls_max_iterations"]):
            out.append(cur); cur *= o["ls_step_reduction_factor"]
            if cur < o["ls_min_step_size"] and i < o["ls_max_iterations"] - 1:
                out.append(o["ls_min_step_size"]); break
        return out or [o["ls_initial_step_size"]]

    # ---- objective (objective.cpp:80-154)
    def err(self, x, t):
        return x - (self.xref_traj[t] if self.xref_traj is not None else self.xref)

    def running_cost(self, x, u, t):
        e = self.err(x, t)
        return float(e @ self.Qdt @ e) + float(u @ self.Rdt @ u)

    def terminal_cost(self, x):
        e = x - self.xref
        return float(e @ self.Qf @ e)

    def total_cost(self, X, U):
        c = 0.0
        for t in range(self.N):
            c += self.running_cost(X[t], U[t], t)
        return c + self.terminal_cost(X[self.N])

    def step(self, x, u, t):
        return discrete_step(self.model, self.integrator, self.dt, x, u, t * self.dt)

    # ---- regularisation (cddp_core.cpp:308-327)
    def reg_up(self):
        self.reg = min(self.reg * self.o["reg_update_factor"], self.o["reg_max_value"])

    def reg_down(self):
        self.reg = max(self.reg / self.o["reg_update_factor"], self.o["reg_min_value"])

    def reg_limit(self):
        return self.reg >= self.o["reg_max_value"]

    # ---- CDDP::setInitialTrajectory + initializeProblemIfNecessary (cddp_core.cpp:272-306)
    def set_initial(self, x0, U0=None, X0=None):
        self.x0 = np.array(x0, float)
        self.X = np.tile(self.x0, (self.N + 1, 1)) if X0 is None else np.array(X0, float).copy()
        self.U = np.zeros((self.N, self.nu)) if U0 is None else np.array(U0, float).copy()
        self.X[0] = self.x0
        self.cost = self.merit = self.inf_pr = self.inf_du = self.inf_comp = INF
        self.reg = self.o["reg_initial_value"]
        self.alpha_pr = self.o["ls_initial_step_size"]; self.alpha_du = 0.0; self.step_norm = 0.0

    def has_term_ineq(self):
        return len(self.term_ineq) > 0

    def term_eq_residual(self, xN):            # ipddp_solver.cpp:155-176
        return np.concatenate([xN[:t.size] - t for _, t in self.term_eq]) if self.term_eq else np.zeros(0)

    def term_eq_jacobian(self):                # :178-201
        H = np.zeros((self.pT, self.nx)); off = 0
        for _, t in self.term_eq:
            H[off:off + t.size, :t.size] = np.eye(t.size); off += t.size
        return H

    # ================================================================ initialize
    def initialize(self):
        N, nx, nu, o = self.N, self.nx, self.nu, self.o
        self.k_u = np.zeros((N, nu)); self.K_u = np.zeros((N, nu, nx)); self.dV = np.zeros(2)
        self.Vx = np.zeros((N + 1, nx)); self.Vxx = np.zeros((N + 1, nx, nx))
        if self.solver == "CLDDP":             # clddp_solver.cpp:28-75 (cold) + computeCost (cddp_solver_base.cpp:416-424)
            self.cost = self.total_cost(self.X, self.U); self.merit = self.cost
            return
        # IPDDP cold start, ipddp_solver.cpp:819-913
        self.Lam = np.zeros((N + 1, nx))
        self.LamT = np.zeros(self.pT); self.dLamT = np.zeros(self.pT)
        X = np.zeros((N + 1, nx)); X[0] = self.x0
        for t in range(N):
            X[t + 1] = self.step(X[t], self.U[t], t)
        self.X = X
        no_cons = (not self.cons) and (not self.term_ineq) and (not self.term_eq)
        self.mu = max(o["tolerance"] / 10.0, o["mu_min_value"]) if no_cons else o["mu_initial"]
        self.reg = o["reg_initial_value"]; self.step_norm = 0.0; self.alpha_pr = 1.0; self.alpha_du = 1.0
        # evaluateTrajectory (:2252-2296) + initializeDualSlackVariables (:2428-2482)
        self.G = self.eval_G(self.X, self.U)
        self.S = np.zeros((N, self.m)); self.Y = np.zeros((N, self.m))
        for t in range(N):
            for i in range(self.m):
                self.S[t, i] = max(o["slack_var_init_scale"], -self.G[t, i] + K_SLACK_INTERIOR_OFFSET)
                self.Y[t, i] = (self.mu * o["dual_var_init_scale"]) / max(self.S[t, i], EPS_SLACK)
        self.cost = self.total_cost(self.X, self.U)
        self.G_T, self.S_T, self.Y_T, self.dS_T, self.dY_T = {}, {}, {}, {}, {}
        for name, A, b in self.term_ineq:       # :889-908
            gT = A @ self.X[N] - b
            s = np.maximum(o["slack_var_init_scale"], -gT + K_SLACK_INTERIOR_OFFSET)
            y = (self.mu * o["dual_var_init_scale"]) / np.maximum(s, EPS_SLACK)
            self.G_T[name], self.S_T[name], self.Y_T[name] = gT, s, y
            self.dS_T[name] = np.zeros_like(s); self.dY_T[name] = np.zeros_like(s)
        self.dS = np.zeros((N, self.m)); self.dY = np.zeros((N, self.m))
        self.k_s = np.zeros((N, self.m)); self.k_y = np.zeros((N, self.m))
        self.K_s = np.zeros((N, self.m, nx)); self.K_y = np.zeros((N, self.m, nx))
        self.reset_filter()
        self.inf_du = 0.0

    def eval_G(self, X, U):
        G = np.zeros((self.N, self.m))
        for t in range(self.N):
            off = 0
            for _, c in self.cons:
                G[t, off:off + c.dim] = c.g(X[t], U[t]); off += c.dim
        return G

    # ---- computeTheta (:2778-2848): constraint-major, then time
    def theta_of(self, G, S, G_T=None, S_T=None, hT=None):
        l2 = self.o["theta_norm"] == "l2"
        total = 0.0; mx = 0.0; off = 0
        for _, c in self.cons:
            for t in range(G.shape[0]):
                r = G[t, off:off + c.dim] + S[t, off:off + c.dim]
                total += float(r @ r) if l2 else float(np.sum(np.abs(r)))
                mx = max(mx, float(np.max(np.abs(r))))
            off += c.dim
        if G_T is not None and S_T is not None:
            for name in sorted(G_T):
                if name not in S_T:
                    continue
                r = G_T[name] + S_T[name]
                total += float(r @ r) if l2 else float(np.sum(np.abs(r)))
                mx = max(mx, float(np.max(np.abs(r))))
        if hT is not None and hT.size > 0:
            total += float(hT @ hT) if l2 else float(np.sum(np.abs(hT)))
            mx = max(mx, float(np.max(np.abs(hT))))
        th = math.sqrt(total) if l2 else total
        return max(th, mx)

    # ---- computeBarrierMerit (:2850-2880)
    def merit_of(self, S, cost, S_T=None, lamT=None, hT=None):
        mer = cost; off = 0
        for _, c in self.cons:
            for t in range(S.shape[0]):
                mer -= self.mu * float(np.sum(np.log(np.maximum(S[t, off:off + c.dim], EPS_SLACK))))
            off += c.dim
        if S_T is not None:
            for name in sorted(S_T):
                mer -= self.mu * float(np.sum(np.log(np.maximum(S_T[name], EPS_SLACK))))
        if lamT is not None and hT is not None and lamT.size == hT.size:
            mer += float(lamT @ hT)
        return mer

    # ---- computePrimalAndComplementarity (:2882-2937)
    def pr_comp_of(self, G, S, Y, mu, G_T=None, S_T=None, Y_T=None, hT=None):
        ipr = 0.0; icomp = 0.0
        if self.m > 0:
            ipr = float(np.max(np.abs(G + S))); icomp = float(np.max(np.abs(Y * S - mu)))
        if G_T is not None and S_T is not None and Y_T is not None:
            for name in G_T:
                if name not in S_T or name not in Y_T:
                    continue
                ipr = max(ipr, float(np.max(np.abs(G_T[name] + S_T[name]))))
                icomp = max(icomp, float(np.max(np.abs(Y_T[name] * S_T[name] - mu))))
        if hT is not None and hT.size > 0:
            ipr = max(ipr, float(np.max(np.abs(hT))))
        return ipr, icomp

    def _term_args(self, X):
        ti = self.has_term_ineq(); te = self.pT > 0
        hT = self.term_eq_residual(X[self.N]) if te else None
        return ti, te, hT

    def reset_filter(self):                    # resetBarrierFilter :2484-2517
        ti, te, hT = self._term_args(self.X)
        self.inf_pr, self.inf_comp = self.pr_comp_of(self.G, self.S, self.Y, self.mu, self.G_T if ti else None,
                                                     self.S_T if ti else None, self.Y_T if ti else None, hT)
        self.merit = self.merit_of(self.S, self.cost, self.S_T if ti else None, self.LamT if te else None, hT)
        self.phi = self.merit
        self.filter_theta = max(self.theta_of(self.G, self.S, self.G_T if ti else None, self.S_T if ti else None, hT), 1e-8)
        self.theta = max(self.filter_theta, max(self.o["theta_0_floor"], 1e-8))
        self.filter = []
        if ti or te:
            self.filter_accept(self.phi, self.filter_theta)

    # ---- interior_point_utils.cpp:79-139
    def filter_accept(self, mf, cv):
        for (fm, fv) in self.filter:
            if fm <= mf and fv <= cv:
                return False
        self.filter = [(fm, fv) for (fm, fv) in self.filter if not (mf <= fm and cv <= fv)]
        self.filter.append((mf, cv))
        return True

    def filter_prune(self):
        if not self.filter:
            return
        bv = min(self.filter, key=lambda p: p[1])          # first minimum, as std::min_element
        bm = min(self.filter, key=lambda p: p[0])
        self.filter = [bv]
        if abs(bm[1] - bv[1]) > 1e-12 or abs(bm[0] - bv[0]) > 1e-12:
            self.filter.append(bm)

    # ================================================================ derivatives
    def lin(self, t):                          # cddp_solver_base.cpp:336-344: A = I + dt f_x, B = dt f_u
        Fx, Fu = self.model.jac(self.X[t], self.U[t], t * self.dt)
        A = self.dt * Fx
        A[np.diag_indices(self.nx)] += 1.0
        return A, self.dt * Fu

    def hess_stack(self, t):                   # cddp_solver_base.cpp:346-356: F_xx_[t][i] = dt * Fxx[i], ... (continuous Hessians)
        Fxx, Fuu, Fux = self.model.hess(self.X[t], self.U[t], t * self.dt)
        return self.dt * Fxx, self.dt * Fuu, self.dt * Fux

    def add_tensor_terms(self, t, w, Q_xx, Q_ux, Q_uu):
        """`for i: Q_xx += w(i) Fxx[i]; Q_ux += w(i) Fux[i]; Q_uu += w(i) Fuu[i]` (ipddp_solver.cpp:1070-1082, 1396-1408)."""
        Fxx, Fuu, Fux = self.hess_stack(t)
        for i in range(self.nx):
            Q_xx = Q_xx + w[i] * Fxx[i]; Q_ux = Q_ux + w[i] * Fux[i]; Q_uu = Q_uu + w[i] * Fuu[i]
        return Q_xx, Q_ux, Q_uu

    def cost_derivs(self, t):
        x, u = self.X[t], self.U[t]
        return 2.0 * self.Qdt @ self.err(x, t), 2.0 * self.Rdt @ u, 2.0 * self.Qdt, 2.0 * self.Rdt, np.zeros((self.nu, self.nx))

    # ================================================================ backward
    def backward(self):
        return self.clddp_backward() if self.solver == "CLDDP" else self.ipddp_backward()

    def clddp_box(self):                       # clddp_solver.cpp:85-86: only a ControlConstraint NAMED "ControlConstraint"
        for name, c in self.cons:
            if name == "ControlConstraint" and type(c) is ControlBox:
                return c
        return None

    def clddp_backward(self):                  # clddp_solver.cpp:79-204
        N, nx, nu, o = self.N, self.nx, self.nu, self.o
        box = self.clddp_box()
        V_x = 2.0 * self.Qf @ (self.X[N] - self.xref); V_xx = 2.0 * self.Qf
        self.Vx[N], self.Vxx[N] = V_x, V_xx
        dV = np.zeros(2); norm_Vx = float(np.sum(np.abs(V_x))); Qu_err = 0.0
        for t in range(N - 1, -1, -1):
            A, B = self.lin(t)
            lx, lu, lxx, luu, lux = self.cost_derivs(t)
            Q_x = lx + A.T @ V_x; Q_u = lu + B.T @ V_x
            Q_xx = lxx + A.T @ V_xx @ A; Q_ux = lux + B.T @ V_xx @ A; Q_uu = luu + B.T @ V_xx @ B
            Q_uu_reg = Q_uu.copy(); Q_uu_reg[np.diag_indices(nu)] += self.reg
            if np.min(np.linalg.eigvals(Q_uu_reg).real) <= 0:     # EigenSolver, :133-140
                return False
            if box is None:
                H = np.linalg.inv(Q_uu_reg)
                k = -H @ Q_u; K = -H @ Q_ux
            else:
                lb = box.lo - self.U[t]; ub = box.up - self.U[t]
                x, status, free, fac = boxqp(Q_uu_reg, Q_u, lb, ub, self.k_u[t].copy(), o)
                if status in ("HESSIAN_NOT_PD", "NO_DESCENT"):
                    return False
                k = x; K = np.zeros((nu, nx))
                idx = np.where(free)[0]
                if idx.size > 0:
                    K[idx, :] = -fac.solve(Q_ux[idx, :])
            self.k_u[t], self.K_u[t] = k, K
            dV += np.array([float(Q_u @ k), 0.5 * float(k @ (Q_uu @ k))])
            V_x = Q_x + K.T @ Q_uu @ k + Q_ux.T @ k + K.T @ Q_u
            V_xx = Q_xx + K.T @ Q_uu @ K + Q_ux.T @ K + K.T @ Q_ux
            V_xx = 0.5 * (V_xx + V_xx.T)
            self.Vx[t], self.Vxx[t] = V_x, V_xx
            norm_Vx += float(np.sum(np.abs(V_x))); Qu_err = max(Qu_err, float(np.max(np.abs(Q_u))))
        self.dV = dV
        sf = o["termination_scaling_max_factor"]
        sf = max(sf, norm_Vx / (N * nx)) / sf
        self.inf_du = Qu_err / sf
        return True

    def ipddp_backward(self):                  # ipddp_solver.cpp:960-1569
        N, nx, nu, m, o, mu = self.N, self.nx, self.nu, self.m, self.o, self.mu
        AB = [self.lin(t) for t in range(N)]
        self.AB = AB
        GJ = []
        for t in range(N):
            Gx = np.zeros((m, nx)); Gu = np.zeros((m, nu)); off = 0
            for _, c in self.cons:
                gx, gu = c.jac(self.X[t], self.U[t]); Gx[off:off + c.dim] = gx; Gu[off:off + c.dim] = gu; off += c.dim
            GJ.append((Gx, Gu))
        V_x = 2.0 * self.Qf @ (self.X[N] - self.xref); V_xx = sym(2.0 * self.Qf)
        dV = np.zeros(2); inf_du = inf_pr = inf_comp = step_norm = 0.0
        ti = self.has_term_ineq(); te = self.pT > 0
        if ti:                                 # :1000-1031
            for name, At, bt in self.term_ineq:
                gT = At @ self.X[N] - bt; self.G_T[name] = gT
                ST, YT = self.S_T[name], self.Y_T[name]
                sig = np.zeros(gT.size); bg = np.zeros(gT.size)
                for i in range(gT.size):
                    ss = max(ST[i], max(mu * 1e-3, EPS_SLACK)); ys = max(YT[i], EPS_DUAL)
                    sig[i] = clip_pos(ys, ss)
                    bg[i] = ys + clip_sgn(ys * gT[i] + mu, ss)
                V_x = V_x + At.T @ bg
                V_xx = sym(V_xx + At.T @ np.diag(sig) @ At)
                inf_pr = max(inf_pr, float(np.max(np.abs(gT + ST))))
                inf_comp = max(inf_comp, float(np.max(np.abs(YT * ST - mu))))
        hT = np.zeros(self.pT); HT = np.zeros((self.pT, nx))
        if te:                                 # :1033-1046
            hT = self.term_eq_residual(self.X[N]); HT = self.term_eq_jacobian()
            inf_pr = max(inf_pr, float(np.max(np.abs(hT))))
            self.dLamT = -hT
        else:
            self.dLamT = np.zeros(0)

        if m == 0 and not ti and not te:       # unconstrained branch :1048-1118
            self.Vx[N], self.Vxx[N] = V_x, V_xx
            for t in range(N - 1, -1, -1):
                A, B = AB[t]
                lx, lu, lxx, luu, lux = self.cost_derivs(t)
                Q_x = lx + A.T @ V_x; Q_u = lu + B.T @ V_x
                Q_xx = lxx + A.T @ V_xx @ A; Q_ux = lux + B.T @ V_xx @ A; Q_uu = luu + B.T @ V_xx @ B
                if not o["use_ilqr"]:
                    Q_xx, Q_ux, Q_uu = self.add_tensor_terms(t, V_x, Q_xx, Q_ux, Q_uu)
                Q_uu = sym(Q_uu); Q_uu[np.diag_indices(nu)] += self.reg
                f = EigenLDLT(Q_uu)
                if not f.ok:
                    return False
                k = -f.solve(Q_u); K = -f.solve(Q_ux)
                self.k_u[t], self.K_u[t] = k, K
                V_x = Q_x + K.T @ Q_u + Q_ux.T @ k + K.T @ Q_uu @ k
                V_xx = sym(Q_xx + K.T @ Q_ux + Q_ux.T @ K + K.T @ Q_uu @ K)
                self.Vx[t], self.Vxx[t] = V_x, V_xx
                dV[0] += float(k @ Q_u); dV[1] += 0.5 * float(k @ (Q_uu @ k))
                inf_du = max(inf_du, float(np.max(np.abs(Q_u)))); step_norm = max(step_norm, float(np.max(np.abs(k))))
            self.dV = dV; self.inf_du = inf_du; self.step_norm = step_norm; self.inf_pr = 0.0; self.inf_comp = 0.0
            return True

        if te:
            return self.ipddp_backward_term_eq(AB, GJ, V_x, V_xx, hT, HT, inf_pr, inf_comp)

        # path / terminal-inequality branch :1355-1568
        self.Vx[N], self.Vxx[N] = V_x, V_xx
        for t in range(N - 1, -1, -1):
            A, B = AB[t]; Q_yx, Q_yu = GJ[t]
            y, s, g = self.Y[t], self.S[t], self.G[t]
            lx, lu, lxx, luu, lux = self.cost_derivs(t)
            Q_x = lx + Q_yx.T @ y + A.T @ V_x
            Q_u = lu + Q_yu.T @ y + B.T @ V_x
            Q_xx = lxx + A.T @ V_xx @ A; Q_ux = lux + B.T @ V_xx @ A; Q_uu = luu + B.T @ V_xx @ B
            if not o["use_ilqr"]:
                Q_xx, Q_ux, Q_uu = self.add_tensor_terms(t, V_x, Q_xx, Q_ux, Q_uu)
            s_safe = np.maximum(s, max(mu * 1e-3, EPS_SLACK))
            YS = np.array([clip_pos(y[i], s_safe[i]) for i in range(m)])
            rp = g + s; rc = y * s - mu; rhat = y * rp - rc
            Q_uu_reg = sym(Q_uu) + Q_yu.T @ np.diag(YS) @ Q_yu
            Q_uu_reg[np.diag_indices(nu)] += self.reg
            f = EigenLDLT(Q_uu_reg)
            if not f.ok:
                return False
            Sir = np.array([clip_sgn(rhat[i], s_safe[i]) for i in range(m)])
            big = np.zeros((nu, 1 + nx))
            big[:, 0] = Q_u + Q_yu.T @ Sir
            big[:, 1:] = Q_ux + Q_yu.T @ np.diag(YS) @ Q_yx
            kK = -f.solve(big)
            k = kK[:, 0].copy(); K = kK[:, 1:].copy()
            self.k_u[t], self.K_u[t] = k, K
            temp = Q_yu @ k
            self.k_y[t] = np.array([clip_sgn(rhat[i] + y[i] * temp[i], s_safe[i]) for i in range(m)])
            self.K_y[t] = np.clip(np.diag(YS) @ (Q_yx + Q_yu @ K), -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO)
            self.k_s[t] = -rp - temp
            self.K_s[t] = -Q_yx - Q_yu @ K
            Q_u = Q_u + Q_yu.T @ Sir; Q_x = Q_x + Q_yx.T @ Sir
            Q_xx = Q_xx + Q_yx.T @ np.diag(YS) @ Q_yx
            Q_ux = Q_ux + Q_yu.T @ np.diag(YS) @ Q_yx
            Q_uu = Q_uu + Q_yu.T @ np.diag(YS) @ Q_yu
            dV[0] += float(k @ Q_u); dV[1] += 0.5 * float(k @ (Q_uu @ k))
            V_x = Q_x + K.T @ Q_u + Q_ux.T @ k + K.T @ Q_uu @ k
            V_xx = sym(Q_xx + K.T @ Q_ux + Q_ux.T @ K + K.T @ Q_uu @ K)
            self.Vx[t], self.Vxx[t] = V_x, V_xx
            inf_du = max(inf_du, float(np.max(np.abs(Q_u))))
            if m > 0:
                inf_pr = max(inf_pr, float(np.max(np.abs(rp)))); inf_comp = max(inf_comp, float(np.max(np.abs(rc))))
            step_norm = max(step_norm, float(np.max(np.abs(k))))
        self.dV = dV
        self.linear_rollout_directions(AB, GJ)
        self.inf_pr, self.inf_du, self.inf_comp, self.step_norm = inf_pr, inf_du, inf_comp, step_norm
        return True

    def linear_rollout_directions(self, AB, GJ, dx0=None):
        """rolloutLinearPolicy with dx0 = 0 (:368-411, 1511-1520), dS / dY (:1522-1532), terminal-inequality
        directions (:1534-1561)."""
        N, nx, mu = self.N, self.nx, self.mu
        dX = np.zeros((N + 1, nx)); dU = np.zeros((N, self.nu))
        for t in range(N):
            dU[t] = self.k_u[t] + self.K_u[t] @ dX[t]
            dX[t + 1] = AB[t][0] @ dX[t] + AB[t][1] @ dU[t] + np.zeros(nx)
        self.dX, self.dU = dX, dU
        for t in range(N):
            self.dS[t] = self.k_s[t] + self.K_s[t] @ dX[t]
            self.dY[t] = np.clip(self.k_y[t] + self.K_y[t] @ dX[t], -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO)
        for name, At, bt in self.term_ineq:
            gT = At @ self.X[N] - bt; ST, YT = self.S_T[name], self.Y_T[name]
            rp = gT + ST; rd = ST * YT - mu
            self.dS_T[name] = -rp - At @ dX[N]
            dYT = np.zeros(gT.size)
            for i in range(gT.size):
                ss = max(ST[i], max(mu * 1e-3, EPS_SLACK))
                ratio = min(max(YT[i] / ss, 0.0), MAX_BARRIER_RATIO)
                aff = min(max(-rd[i] / ss, -MAX_BARRIER_RATIO), MAX_BARRIER_RATIO)
                dYT[i] = min(max(aff - ratio * self.dS_T[name][i], -MAX_BARRIER_RATIO), MAX_BARRIER_RATIO)
            self.dY_T[name] = dYT

    # ---- terminal-equality branch, filled in by cddp_twin_te.py (kept separate: it is the longest piece)
    def ipddp_backward_term_eq(self, AB, GJ, V_x, V_xx, hT, HT, inf_pr, inf_comp):
        from cddp_twin_te import backward_term_eq
        return backward_term_eq(self, AB, GJ, V_x, V_xx, hT, HT, inf_pr, inf_comp)

    # ================================================================ forward
    def forward(self, alpha):
        return self.clddp_forward(alpha) if self.solver == "CLDDP" else self.ipddp_forward(alpha)

    def clddp_forward(self, a):                # clddp_solver.cpp:215-262
        N = self.N; box = self.clddp_box()
        r = dict(success=False, cost=INF, merit=INF, alpha_pr=a, alpha_du=1.0, alpha=a)
        X = self.X.copy(); U = self.U.copy(); X[0] = self.x0
        J = 0.0
        for t in range(N):
            dx = X[t] - self.X[t]
            U[t] = U[t] + a * self.k_u[t] + self.K_u[t] @ dx
            if box is not None:
                U[t] = np.minimum(np.maximum(U[t], box.lo), box.up)
            J += self.running_cost(X[t], U[t], t)
            X[t + 1] = self.step(X[t], U[t], t)
        J += self.terminal_cost(X[N])
        dJ = self.cost - J
        expected = -a * (self.dV[0] + 0.5 * a * self.dV[1])
        ratio = dJ / expected if expected > 0.0 else math.copysign(1.0, dJ)
        r.update(success=ratio > self.o["filter_armijo_constant"], cost=J, merit=J, X=X, U=U)
        return r

    def max_step_sizes(self):                  # computeMaxStepSizes :2939-2988
        tau = max(self.o["min_fraction_to_boundary"], 1.0 - self.mu)
        apr = adu = 1.0
        off = 0
        for _, c in self.cons:
            for t in range(self.N):
                for i in range(off, off + c.dim):
                    if self.dS[t, i] < 0.0:
                        apr = min(apr, -tau * self.S[t, i] / self.dS[t, i])
                    if self.dY[t, i] < 0.0:
                        adu = min(adu, -tau * self.Y[t, i] / self.dY[t, i])
            off += c.dim
        for name, _, _ in self.term_ineq:
            s, y, ds, dy = self.S_T[name], self.Y_T[name], self.dS_T[name], self.dY_T[name]
            for i in range(s.size):
                if ds[i] < 0.0:
                    apr = min(apr, -tau * s[i] / ds[i])
                if dy[i] < 0.0:
                    adu = min(adu, -tau * y[i] / dy[i])
        return min(max(apr, 0.0), 1.0), min(max(adu, 0.0), 1.0)

    def ipddp_forward(self, alpha):            # ipddp_solver.cpp:1571-1876
        N, nx, m, o, mu = self.N, self.nx, self.m, self.o, self.mu
        ti = self.has_term_ineq(); te = self.pT > 0
        apm, adm = self.max_step_sizes()
        r = dict(success=False, cost=self.cost, merit=self.phi, theta=self.theta, alpha=alpha, inf_pr=0.0, inf_comp=0.0)
        tau = 1.0 if (not self.cons and not ti) else max(o["min_fraction_to_boundary"], 1.0 - mu)
        a_pr = min(alpha, apm); a_du = min(alpha, adm)
        r["alpha_pr"], r["alpha_du"] = a_pr, a_du
        X = np.zeros((N + 1, nx)); U = np.zeros((N, self.nu)); X[0] = self.x0
        Lam = self.Lam.copy(); S = self.S.copy(); Y = self.Y.copy()
        S_T = {k: v.copy() for k, v in self.S_T.items()}; Y_T = {k: v.copy() for k, v in self.Y_T.items()}
        G_T = {k: v.copy() for k, v in self.G_T.items()}
        LamT = self.LamT.copy()
        for t in range(N):
            dx = X[t] - self.X[t]
            Lam[t] = self.Lam[t] + a_pr * self.Vx[t] + self.Vxx[t] @ dx
            if not np.all(np.isfinite(Lam[t])):
                return r
            off = 0
            for _, c in self.cons:
                sl = slice(off, off + c.dim); off += c.dim
                s_new = self.S[t, sl] + a_pr * self.k_s[t, sl] + self.K_s[t, sl] @ dx
                s_min = (1.0 - tau) * self.S[t, sl]
                y_new = self.Y[t, sl] + a_du * self.k_y[t, sl] + self.K_y[t, sl] @ dx
                y_min = (1.0 - tau) * self.Y[t, sl]
                if np.any((s_new < s_min) | (y_new < y_min)):
                    return r
                if not (np.all(np.isfinite(s_new)) and np.all(np.isfinite(y_new))):
                    return r
                S[t, sl] = s_new; Y[t, sl] = y_new
            U[t] = self.U[t] + a_pr * self.k_u[t] + self.K_u[t] @ dx
            X[t + 1] = self.step(X[t], U[t], t)
            if not (np.all(np.isfinite(X[t + 1])) and np.all(np.isfinite(U[t]))):
                return r
        dxN = X[N] - self.X[N]
        Lam[N] = self.Lam[N] + a_pr * self.Vx[N] + self.Vxx[N] @ dxN
        if not np.all(np.isfinite(Lam[N])):
            return r
        if ti:                                 # :1672-1722
            for name, At, bt in self.term_ineq:
                g0 = At @ self.X[N] - bt
                ST0, YT0 = self.S_T[name], self.Y_T[name]
                k_sT = -(g0 + ST0); K_sT = -At
                S_T[name] = ST0 + a_pr * k_sT + K_sT @ dxN
                Yt = YT0.copy()
                for i in range(g0.size):
                    ss = max(ST0[i], max(mu * 1e-3, EPS_SLACK))
                    rd = YT0[i] * ST0[i] - mu
                    ratio = clip_pos(YT0[i], ss)
                    Ky = -(ratio * K_sT[i])
                    ky = clip_sgn(-rd - YT0[i] * k_sT[i], ss)
                    Yt[i] = YT0[i] + a_du * ky + float(Ky @ dxN)
                Y_T[name] = Yt
                s_floor = np.maximum((1.0 - tau) * ST0, max(mu * 1e-3, EPS_SLACK))
                if (np.any(S_T[name] < s_floor) or np.any(Y_T[name] < (1.0 - tau) * YT0)
                        or not np.all(np.isfinite(S_T[name])) or not np.all(np.isfinite(Y_T[name]))):
                    return r
        if te:
            LamT = self.LamT + a_pr * self.dLamT
            if not np.all(np.isfinite(LamT)):
                return r
        cost_new = 0.0
        for t in range(N):
            cost_new += self.running_cost(X[t], U[t], t)
        cost_new += self.terminal_cost(X[N])
        G = self.eval_G(X, U)
        hT = None
        if ti:
            for name, At, bt in self.term_ineq:
                G_T[name] = At @ X[N] - bt
        if te:
            hT = self.term_eq_residual(X[N])
        phi = self.merit_of(S, cost_new, S_T if ti else None, LamT if te else None, hT)
        theta = self.theta_of(G, S, G_T if ti else None, S_T if ti else None, hT)
        ipr, icomp = self.pr_comp_of(G, S, Y, mu, G_T if ti else None, S_T if ti else None, Y_T if ti else None, hT)
        if not all(math.isfinite(v) for v in (phi, theta, ipr, icomp)):
            return r
        acc = False
        if not self.cons and not ti and not te:
            dJ = self.cost - cost_new
            expected = -a_pr * (self.dV[0] + 0.5 * a_pr * self.dV[1])
            ratio = dJ / expected if expected > 0.0 else math.copysign(1.0, dJ)
            acc = ratio > 1e-6
        else:
            exp_impr = a_pr * self.dV[0]
            cv_old = self.filter[-1][1] if self.filter else 0.0
            high_ref = self.filter_theta if not self.filter else cv_old
            merit_old = self.merit
            if theta > o["filter_max_violation_threshold"]:
                acc = theta < (1 - o["filter_violation_acceptance_threshold"]) * high_ref
            elif max(theta, cv_old) < o["filter_min_violation_for_armijo_check"] and exp_impr < 0:
                acc = phi < merit_old + o["filter_armijo_constant"] * exp_impr
            else:
                acc = (phi < merit_old - o["filter_merit_acceptance_threshold"] * theta
                       or theta < (1 - o["filter_violation_acceptance_threshold"]) * cv_old)
        if not acc:
            return r
        r.update(success=True, cost=cost_new, merit=phi, theta=theta, inf_pr=ipr, inf_comp=icomp, X=X, U=U, S=S, Y=Y, G=G,
                 Lam=Lam, S_T=S_T, Y_T=Y_T, G_T=G_T, LamT=LamT)
        return r

    # ================================================================ outer loop (cddp_solver_base.cpp:29-186)
    def scaled_inf_du(self):                   # computeScaledDualInfeasibility :2725-2776
        v = self.inf_du
        if not self.o["check_state_stationarity"]:
            return v
        ss = 0.0
        for t in range(self.N):
            off = 0
            for _, c in self.cons:
                gx, _ = c.jac(self.X_lin[t], self.U_lin[t])
                ss = max(ss, float(np.max(np.abs(gx.T @ self.Y[t, off:off + c.dim])))); off += c.dim
        return max(v, ss)

    def record(self):
        self.history.append([self.cost, self.merit, self.alpha_pr, self.alpha_du, self.inf_du, self.inf_pr, self.inf_comp,
                             self.mu if self.solver == "IPDDP" else 0.0, self.reg])

    def no_barrier(self):
        return (not self.cons) and (not self.term_ineq)

    def early_convergence(self):
        o = self.o
        if self.solver == "CLDDP":             # clddp_solver.cpp:206-213
            return self.inf_du < o["tolerance"]
        sdu = self.scaled_inf_du()             # ipddp_solver.cpp:925-958
        if self.no_barrier():
            return self.inf_pr < o["tolerance"] and sdu < o["tolerance"]
        tol = max(o["tolerance"], o["barrier_tol_mult"] * self.mu)
        return (self.inf_pr < tol and sdu < tol and self.inf_comp < tol
                and abs(self.alpha_pr) * self.step_norm < o["tolerance"] * 10.0)

    def line_search(self):                     # performForwardPass, cddp_solver_base.cpp:248-317
        best = dict(success=False, cost=INF, merit=INF)
        self.n_forward_last = 0
        for a in self.alphas:
            r = self.forward(a); self.n_forward_last += 1
            if self.o["enable_parallel"]:
                if r["success"] and r["merit"] < best["merit"]:
                    best = r
            elif r["success"]:
                best = r; break
        if self.o["enable_parallel"]:
            self.n_forward_last = (self.alphas.index(best["alpha"]) + 1) if best["success"] else len(self.alphas)
        return best

    def apply(self, r):                        # applyForwardPassResult: base :190-198 + ipddp_solver.cpp:1878-1951
        self.X_lin, self.U_lin = self.X, self.U                   # G_x_ keeps the Jacobians of the last backward pass
        self.X, self.U, self.cost, self.merit = r["X"], r["U"], r["cost"], r["merit"]
        self.alpha_pr, self.alpha_du = r["alpha_pr"], r["alpha_du"]
        if self.solver == "CLDDP":
            return
        self.Y, self.S, self.G, self.Lam = r["Y"], r["S"], r["G"], r["Lam"]
        if self.has_term_ineq():
            self.S_T, self.Y_T, self.G_T = r["S_T"], r["Y_T"], r["G_T"]
        if self.pT > 0:
            self.LamT = r["LamT"]
        self.inf_pr, self.inf_comp = r["inf_pr"], r["inf_comp"]
        self.phi = r["merit"]; self.filter_theta = r["theta"]; self.theta = r["theta"]
        self.update_barrier()

    def update_barrier(self):                  # updateBarrierParameters(true) :2548-2660
        o = self.o
        sdu = self.scaled_inf_du(); mu_old = self.mu
        if self.no_barrier():
            pass
        elif o["barrier_strategy"] == "ADAPTIVE":
            kkt = max(self.inf_pr, sdu, self.inf_comp)
            if kkt <= max(o["mu_update_factor"] * self.mu, 2.0 * self.mu):
                factor = o["mu_update_factor"]
                if self.mu > 1e-20:
                    ratio = kkt / max(self.mu, 1e-20)
                    if ratio < 0.01:
                        factor = 0.1 * o["mu_update_factor"]
                    elif ratio < 0.1:
                        factor = 0.3 * o["mu_update_factor"]
                    elif ratio < 0.5:
                        factor = 0.6 * o["mu_update_factor"]
                self.mu = max(min(factor * self.mu, self.mu ** o["mu_update_power"]), max(o["mu_min_value"], o["tolerance"] / 100.0))
        else:
            kkt = max(self.inf_pr, sdu * o["barrier_update_dual_weight"], self.inf_comp)
            if kkt <= o["mu_kappa_epsilon"] * self.mu:
                self.mu = max(o["mu_min_value"], min(o["mu_update_factor"] * self.mu, self.mu ** o["mu_update_power"]))
        ti, te, hT = self._term_args(self.X)
        ftheta = max(self.theta_of(self.G, self.S, self.G_T if ti else None, self.S_T if ti else None, hT), 1e-8)
        if self.mu < mu_old and self.mu > 0.0:
            self.filter = []
            if ti or te:
                self.filter_accept(self.phi, ftheta)
        else:
            self.filter_accept(self.phi, ftheta)
            if len(self.filter) > o["max_filter_size"]:
                self.filter_prune()
        self.inf_pr, self.inf_comp = self.pr_comp_of(self.G, self.S, self.Y, self.mu, self.G_T if ti else None,
                                                     self.S_T if ti else None, self.Y_T if ti else None, hT)
        self.merit = self.merit_of(self.S, self.cost, self.S_T if ti else None, self.LamT if te else None, hT)
        self.phi = self.merit
        self.filter_theta = ftheta
        self.theta = max(ftheta, max(o["theta_0_floor"], 1e-8))

    def check_convergence(self, dJ, it):
        o = self.o
        if self.solver == "CLDDP":             # clddp_solver.cpp:264-277
            if self.inf_du < o["tolerance"]:
                return 1
            if dJ > 0.0 and dJ < o["acceptable_tolerance"]:
                return 2
            return 0
        sdu = self.scaled_inf_du(); scomp = self.inf_comp   # ipddp_solver.cpp:1953-2025
        if self.no_barrier():
            if self.inf_pr < o["tolerance"] and sdu < o["tolerance"]:
                return 1
            if o["acceptable_tolerance"] > 0.0:
                sq = math.sqrt(o["acceptable_tolerance"])
                acc = self.inf_pr < sq and sdu < sq and it > 50
                if dJ > 0.0:
                    acc = acc or (dJ < o["acceptable_tolerance"] and it > 50 and self.inf_pr < sq and sdu < sq)
                if acc:
                    return 2
            return 0
        tol = max(o["tolerance"], o["barrier_tol_mult"] * self.mu)
        if self.inf_pr < tol and sdu < tol and scomp < tol and self.step_norm < o["tolerance"] * 10.0:
            return 1
        if o["acceptable_tolerance"] > 0.0:
            at = math.sqrt(o["acceptable_tolerance"])
            bat = max(o["mu_min_value"] * 100.0, o["tolerance"] / 10.0)
            akkt = self.inf_pr < at and sdu < at and scomp < at
            bpc = self.mu <= bat
            acc = akkt and bpc and it > 10 and abs(dJ) < o["acceptable_tolerance"]
            acc = acc or (akkt and bpc and it >= 1 and self.step_norm < o["tolerance"] * 10.0 and self.inf_pr < 1e-4)
            if acc:
                return 2
        return 0

    def forward_failure(self):
        """handleForwardPassFailure: base cddp_solver_base.cpp:206-218, IPDDP ipddp_solver.cpp:2037-2082.
        Returns the terminating status or 0."""
        o = self.o
        self.reg_up()
        if self.solver == "CLDDP":
            return 4 if self.reg_limit() else 0
        nb = self.no_barrier()
        if (not nb) and self.pT > 0:
            self.reg_up()
        if self.reg_limit():
            sdu = self.scaled_inf_du()
            base = math.sqrt(max(o["acceptable_tolerance"], o["tolerance"]))
            at = base if nb else max(base, o["barrier_tol_mult"] * self.mu)
            acc = (o["acceptable_tolerance"] > 0.0 and self.inf_pr < at and sdu < at and (nb or self.inf_comp < at))
            return 2 if acc else 4
        return 0

    def solve(self):
        self.history = []; self.n_backward = 0; self.n_forward = 0
        self.initialize()
        self.X_lin, self.U_lin = self.X, self.U
        self.record()
        it = 0; status = 3
        while it < self.o["max_iterations"]:
            it += 1
            ok = False
            while not ok:
                self.n_backward += 1
                self.X_lin, self.U_lin = self.X, self.U
                ok = self.backward()
                if not ok:
                    self.reg_up()
                    if self.reg_limit():
                        status = 4; break
            if not ok:
                break
            if self.early_convergence():
                status = 1; self.record(); break
            best = self.line_search(); self.n_forward += self.n_forward_last
            if best["success"]:
                dJ = self.cost - best["cost"]
                self.apply(best)
                self.record()
                self.reg_down()
                c = self.check_convergence(dJ, it)
                if c:
                    status = c; break
            else:
                s = self.forward_failure()
                if s:
                    status = s; break
        self.iterations = it; self.status = status
        return dict(iterations=it, status=status, final_objective=self.cost, n_backward=self.n_backward, n_forward=self.n_forward)
