"""ctypes binding of the CPU oracle (oracle/cddp_oracle.cpp) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this module; the product package
(cddp-cpp_amd/) neither imports it nor knows where the oracle libraries are.  `attach(api)` hangs the oracle entry
points (Oracle, oracle_solve_batch, ...) on the product's harness module so the tests address both through one
namespace.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
ORACLE_LIB_PATH = os.environ.get("CDDP_ORACLE_LIB") or os.path.join(_HERE, "_build", "libcddp_oracle.so")   # override: the sanitizer build (make -C oracle sanitize)
ORACLE_FAST_LIB_PATH = os.path.join(_HERE, "_build", "libcddp_oracle_fast.so")


def _api():
    return sys.modules["cddp_cpp_amd_pyapi"]


_oracle_libs = {}


_trig_mode = [0]


def load_oracle(fast=False):
    path = ORACLE_FAST_LIB_PATH if fast else ORACLE_LIB_PATH
    if path in _oracle_libs:
        return _oracle_libs[path]
    if not os.path.exists(path):
        raise RuntimeError("oracle library missing: %s (run __graft_entry__.build() or make -C oracle)" % path)
    lib = C.CDLL(path)
    lib.cddp_oracle_create.restype = C.c_void_p
    lib.cddp_oracle_create.argtypes = [C.POINTER(_api().ProblemStruct)]
    lib.cddp_oracle_destroy.argtypes = [C.c_void_p]
    for name in ["cddp_oracle_filter_theta", "cddp_oracle_filter_back_violation", "cddp_oracle_scaled_inf_du",
                 "cddp_oracle_get_mu", "cddp_oracle_cost"]:
        getattr(lib, name).restype = C.c_double
    _oracle_libs[path] = lib
    if _trig_mode[0]:
        lib.cddp_oracle_set_trig_mode(_trig_mode[0])   # a library loaded after set_trig_mode() joins the current mode
    return lib


class Oracle:
    def __init__(self, problem, fast=False):
        self.lib = load_oracle(fast)
        self.p = problem
        self.h = C.c_void_p(self.lib.cddp_oracle_create(C.byref(problem.c)))
        self.m = self.lib.cddp_oracle_dual_dim(self.h)

    def __del__(self):
        try:
            self.lib.cddp_oracle_destroy(self.h)
        except Exception:
            pass

    def set_initial(self, x0, U0=None, X0=None):
        x0 = _api()._arr(x0); U0 = _api()._arr(U0) if U0 is not None else None; X0 = _api()._arr(X0) if X0 is not None else None
        self.lib.cddp_oracle_set_initial(self.h, _api()._ptr(x0), _api()._ptr(U0), _api()._ptr(X0))

    def initialize(self):
        self.lib.cddp_oracle_initialize(self.h)

    # ---- warm-start plumbing (reference: options.warm_start, IPDDPSolverTestAccess, setInitialState/Trajectory)
    def set_warm_start(self, flag=True):
        self.lib.cddp_oracle_set_warm_start(self.h, 1 if flag else 0)

    def set_path_interior(self, s_val, y_val):
        self.lib.cddp_oracle_set_path_interior(self.h, C.c_double(s_val), C.c_double(y_val))

    def set_terminal_interior(self, s_val, y_val):
        self.lib.cddp_oracle_set_terminal_interior(self.h, C.c_double(s_val), C.c_double(y_val))

    def set_terminal_eq_multiplier(self, lam):
        lam = _api()._arr(lam)
        self.lib.cddp_oracle_set_terminal_eq_multiplier(self.h, _api()._ptr(lam))

    def update_initial(self, x0, U0=None):
        x0 = _api()._arr(x0); U0 = _api()._arr(U0) if U0 is not None else None
        self.lib.cddp_oracle_update_initial(self.h, _api()._ptr(x0), _api()._ptr(U0))

    def backward(self, retry=True):
        return self.lib.cddp_oracle_backward(self.h, 1 if retry else 0)

    def forward(self, alpha):
        t = np.zeros(1, dtype=_api().TRIAL_DTYPE)
        self.lib.cddp_oracle_forward(self.h, C.c_double(alpha), t.ctypes.data_as(C.c_void_p))
        return t[0]

    def solve(self):
        r = np.zeros(1, dtype=_api().RESULT_DTYPE)
        self.lib.cddp_oracle_solve(self.h, r.ctypes.data_as(C.c_void_p))
        return r[0]

    def result(self):
        r = np.zeros(1, dtype=_api().RESULT_DTYPE)
        self.lib.cddp_oracle_get_result(self.h, r.ctypes.data_as(C.c_void_p))
        return r[0]

    def alphas(self):
        a = np.zeros(64)
        n = self.lib.cddp_oracle_num_alphas(self.h, _api()._ptr(a), 64)
        return a[:n].copy()

    def trajectory(self):
        X = np.zeros((self.p.N + 1, self.p.nx)); U = np.zeros((self.p.N, self.p.nu))
        self.lib.cddp_oracle_get_trajectory(self.h, _api()._ptr(X), _api()._ptr(U))
        return X, U

    def gains(self):
        K = np.zeros((self.p.N, self.p.nu, self.p.nx)); k = np.zeros((self.p.N, self.p.nu))
        self.lib.cddp_oracle_get_gains(self.h, _api()._ptr(K), _api()._ptr(k))
        return K, k

    def value(self):
        Vx = np.zeros((self.p.N + 1, self.p.nx)); Vxx = np.zeros((self.p.N + 1, self.p.nx, self.p.nx))
        self.lib.cddp_oracle_get_value(self.h, _api()._ptr(Vx), _api()._ptr(Vxx))
        return Vx, Vxx

    def duals(self):
        S = np.zeros((self.p.N, self.m)); Y = np.zeros((self.p.N, self.m)); G = np.zeros((self.p.N, self.m))
        if self.m:
            self.lib.cddp_oracle_get_duals(self.h, _api()._ptr(S), _api()._ptr(Y), _api()._ptr(G))
        return S, Y, G

    def costates(self):
        """Costate trajectory Lambda of the current iterate (rows, nx): N + 1 rows under IPDDP, N under MSIPDDP."""
        rows = self.lib.cddp_oracle_get_costates(self.h, None)
        L = np.zeros((rows, self.p.nx))
        if rows:
            self.lib.cddp_oracle_get_costates(self.h, _api()._ptr(L))
        return L

    def terminal(self):
        dims = np.zeros(2, dtype=np.int32)
        self.lib.cddp_oracle_get_terminal(self.h, None, None, None, None, dims.ctypes.data_as(C.POINTER(C.c_int32)))
        mT, pT = int(dims[0]), int(dims[1])
        S = np.zeros(mT); Y = np.zeros(mT); G = np.zeros(mT); L = np.zeros(pT)
        self.lib.cddp_oracle_get_terminal(self.h, _api()._ptr(S), _api()._ptr(Y), _api()._ptr(G), _api()._ptr(L), None)
        return S, Y, G, L

    def backward_scalars(self):
        dV = np.zeros(2); reg = np.zeros(1)
        self.lib.cddp_oracle_get_backward_scalars(self.h, _api()._ptr(dV), _api()._ptr(reg))
        return dV, reg[0]

    def history(self):
        cap = self.p.options.max_iterations + 2
        h = np.zeros((cap, 9))
        n = self.lib.cddp_oracle_get_history(self.h, _api()._ptr(h), cap)
        return h[:n].copy()

    def dynamics(self, x, u, time=0.0):
        x = _api()._arr(x); u = _api()._arr(u)
        xd = np.zeros(self.p.nx); xn = np.zeros(self.p.nx)
        Fx = np.zeros((self.p.nx, self.p.nx)); Fu = np.zeros((self.p.nx, self.p.nu))
        self.lib.cddp_oracle_dynamics(self.h, _api()._ptr(x), _api()._ptr(u), C.c_double(time), _api()._ptr(xd), _api()._ptr(xn), _api()._ptr(Fx), _api()._ptr(Fu))
        return xd, xn, Fx, Fu

    def hessians(self, x, u):
        x = _api()._arr(x); u = _api()._arr(u); nx, nu = self.p.nx, self.p.nu
        Fxx = np.zeros((nx, nx, nx)); Fuu = np.zeros((nx, nu, nu)); Fux = np.zeros((nx, nu, nx))
        ok = self.lib.cddp_oracle_hessians(self.h, _api()._ptr(x), _api()._ptr(u), _api()._ptr(Fxx), _api()._ptr(Fuu), _api()._ptr(Fux))
        return (Fxx, Fuu, Fux) if ok else None

    def constraint_eval(self, x, u):
        x = _api()._arr(x); u = _api()._arr(u)
        g = np.zeros(self.m); gx = np.zeros((self.m, self.p.nx)); gu = np.zeros((self.m, self.p.nu))
        self.lib.cddp_oracle_constraint_eval(self.h, _api()._ptr(x), _api()._ptr(u), _api()._ptr(g), _api()._ptr(gx), _api()._ptr(gu))
        return g, gx, gu

    def cost(self, X, U):
        X = _api()._arr(X); U = _api()._arr(U)
        return self.lib.cddp_oracle_cost(self.h, _api()._ptr(X), _api()._ptr(U))


def oracle_solve_batch(problem, x0, U0=None, X0=None, n_threads=1, fast=False, want_traj=True):
    lib = load_oracle(fast)
    x0 = _api()._arr(x0); B = x0.shape[0]
    U0 = _api()._arr(U0) if U0 is not None else None; X0 = _api()._arr(X0) if X0 is not None else None
    res = np.zeros(B, dtype=_api().RESULT_DTYPE)
    X = np.zeros((B, problem.N + 1, problem.nx)) if want_traj else None
    U = np.zeros((B, problem.N, problem.nu)) if want_traj else None
    K = np.zeros((B, problem.N, problem.nu, problem.nx)) if want_traj else None
    ms = C.c_double(0.0)
    lib.cddp_oracle_solve_batch(C.byref(problem.c), B, _api()._ptr(x0), _api()._ptr(U0), _api()._ptr(X0), n_threads,
                                res.ctypes.data_as(C.c_void_p), _api()._ptr(X), _api()._ptr(U), _api()._ptr(K), C.byref(ms))
    return res, X, U, K, ms.value


def oracle_boxqp(H, g, lower, upper, x0=None, options=None):
    lib = load_oracle()
    o = options if options is not None else _api().default_options()
    H = _api()._arr(H); g = _api()._arr(g); lo = _api()._arr(lower); up = _api()._arr(upper); n = g.size
    x0a = _api()._arr(x0) if x0 is not None else None
    x = np.zeros(n); free = np.zeros(n, dtype=np.int32); it = C.c_int(0); fc = C.c_int(0)
    st = lib.cddp_oracle_boxqp(C.byref(o), n, _api()._ptr(H), _api()._ptr(g), _api()._ptr(lo), _api()._ptr(up), _api()._ptr(x0a), _api()._ptr(x),
                               free.ctypes.data_as(C.POINTER(C.c_int)), C.byref(it), C.byref(fc))
    return x, st, free, it.value, fc.value


def oracle_ldlt_solve(A, B):
    lib = load_oracle()
    A = _api()._arr(A); B = _api()._arr(B); n = A.shape[0]; B2 = B.reshape(n, -1); X = np.zeros_like(B2)
    ok = lib.cddp_oracle_ldlt_solve(n, B2.shape[1], _api()._ptr(A), _api()._ptr(B2), _api()._ptr(X))
    return X.reshape(B.shape), bool(ok)




def set_trig_mode(mode):
    """0 = glibc (the reference's arithmetic: golden fixtures, twin comparisons), 1 = the straight-line sin / cos / log / pow of
    cddp-cpp_amd/csrc/dev_trig.hpp that the HIP library evaluates (models.hpp::trig_mode).  Applied to both oracle builds;
    returns the previous mode.  Process-global, like the noise knobs."""
    prev = _trig_mode[0]
    for fast in (False, True):
        path = ORACLE_FAST_LIB_PATH if fast else ORACLE_LIB_PATH
        if fast and not os.path.exists(path):
            continue
        load_oracle(fast).cddp_oracle_set_trig_mode(int(mode))
    _trig_mode[0] = int(mode)
    return prev


def set_failing_alphas(mask):
    """Test hook: the forward pass of every alpha index whose bit is set is evaluated and discarded (cddp_oracle.cpp::performForwardPass;
    the reference's parallel rule with a throwing forward pass, cddp_solver_base.cpp:280-296).  0 = off.  Process-global, both builds."""
    for fast in (False, True):
        path = ORACLE_FAST_LIB_PATH if fast else ORACLE_LIB_PATH
        if fast and not os.path.exists(path):
            continue
        load_oracle(fast).cddp_oracle_set_failing_alphas(int(mask))


def set_assoc_mode(mode):
    """Summation-order model (linalg.hpp::assoc_mode): 0 = every sum serial (default, the order the HIP kernels keep), 1 = the association
    Eigen 3.4's SSE2 kernels give the reference's dot products, norms and transposed matrix-vector products.  Process-global, both builds."""
    for fast in (False, True):
        path = ORACLE_FAST_LIB_PATH if fast else ORACLE_LIB_PATH
        if fast and not os.path.exists(path):
            continue
        load_oracle(fast).cddp_oracle_set_assoc_mode(int(mode))


class shared_trig:
    """Context manager: the oracle evaluates sin / cos / log / pow with the HIP library's routines (models.hpp::trig_mode 1) inside
    the block and returns to the previous mode afterwards (tests/conftest.py keeps mode 1 on for every `-m gpu` test)."""

    def __init__(self, fast=False):
        self.prev = 0

    def __enter__(self):
        self.prev = set_trig_mode(1)
        return self

    def __exit__(self, *exc):
        set_trig_mode(self.prev)
        return False


def attach(api):
    """Expose the oracle entry points on the harness module `api` (cddp-cpp_amd/pyapi.py)."""
    mod = sys.modules[__name__]
    for name in ("ORACLE_LIB_PATH", "ORACLE_FAST_LIB_PATH", "load_oracle", "Oracle", "oracle_solve_batch", "oracle_boxqp",
                 "oracle_ldlt_solve", "shared_trig", "set_trig_mode", "set_failing_alphas", "set_assoc_mode"):
        setattr(api, name, getattr(mod, name))
    return api
