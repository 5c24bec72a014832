"""ctypes binding of the C-ABI in include/cddp_hip.h.

Builds `cddp_hip_problem` descriptors (the POD twin of cddp::CDDP, reference
include/cddp-cpp/cddp_core/cddp_core.hpp:215-423) and calls cddp-cpp_amd/lib/libcddp_hip.so (`HipBatchSolver`; fails
loudly if the library or a GPU is missing).  It contains no solver arithmetic and knows nothing about the CPU checker:
the tests / bench.py's cpu_baseline hang that on this namespace themselves.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
HIP_LIB_PATH = os.environ.get("CDDP_HIP_LIB") or os.path.join(_HERE, "lib", "libcddp_hip.so")   # override: kernel experiments

ABI_VERSION = 5          # CDDP_HIP_ABI_VERSION of include/cddp_hip.h
MAX_MODEL_PARAMS = 24
NAME_LEN = 48

# enums (include/cddp_hip.h)
MODEL_PENDULUM, MODEL_CARTPOLE, MODEL_UNICYCLE, MODEL_LTI = 0, 1, 2, 3
MODEL_QUADROTOR, MODEL_MANIPULATOR, MODEL_QUADROTOR_EULER12, MODEL_MANIPULATOR7 = 4, 5, 6, 7
MODEL_BICYCLE, MODEL_CAR, MODEL_HCW = 8, 9, 10
EULER, HEUN, RK3, RK4 = 0, 1, 2, 3
SOLVER_CLDDP, SOLVER_IPDDP, SOLVER_LOGDDP, SOLVER_MSIPDDP = 0, 1, 2, 3
CON_CONTROL_BOX, CON_STATE_BOX, CON_BALL, CON_LINEAR = 0, 1, 2, 3
CON_SOC, CON_THRUST, CON_MAX_THRUST = 4, 5, 6
TERM_EQUALITY, TERM_INEQUALITY = 0, 1
STATUS_RUNNING, STATUS_OPTIMAL, STATUS_ACCEPTABLE, STATUS_MAX_ITERATIONS, STATUS_REG_LIMIT, STATUS_MAX_CPU_TIME, STATUS_REG_LIMIT_CONVERGED = range(7)
STATUS_STRINGS = [
    "Running", "OptimalSolutionFound", "AcceptableSolutionFound", "MaxIterationsReached",
    "RegularizationLimitReached_NotConverged", "MaxCpuTimeReached", "RegularizationLimitReached_Converged",
]

_dp = C.POINTER(C.c_double)


class Options(C.Structure):
    _fields_ = [
        ("tolerance", C.c_double), ("acceptable_tolerance", C.c_double),
        ("max_iterations", C.c_int32), ("use_ilqr", C.c_int32), ("enable_parallel", C.c_int32),
        ("return_iteration_info", C.c_int32), ("warm_start", C.c_int32), ("_pad0", C.c_int32),
        ("termination_scaling_max_factor", C.c_double),
        ("ls_max_iterations", C.c_int32), ("_pad1", C.c_int32),
        ("ls_initial_step_size", C.c_double), ("ls_min_step_size", C.c_double),
        ("ls_step_reduction_factor", C.c_double),
        ("reg_initial_value", C.c_double), ("reg_update_factor", C.c_double),
        ("reg_max_value", C.c_double), ("reg_min_value", C.c_double),
        ("boxqp_max_iterations", C.c_int32), ("_pad2", C.c_int32),
        ("boxqp_min_gradient_norm", C.c_double), ("boxqp_min_relative_improvement", C.c_double),
        ("boxqp_step_decrease_factor", C.c_double), ("boxqp_min_step_size", C.c_double),
        ("boxqp_armijo_constant", C.c_double),
        ("filter_merit_acceptance_threshold", C.c_double),
        ("filter_violation_acceptance_threshold", C.c_double),
        ("filter_max_violation_threshold", C.c_double),
        ("filter_min_violation_for_armijo_check", C.c_double),
        ("filter_armijo_constant", C.c_double),
        ("ipddp_dual_var_init_scale", C.c_double), ("ipddp_slack_var_init_scale", C.c_double),
        ("ipddp_barrier_tol_mult", C.c_double), ("ipddp_barrier_update_dual_weight", C.c_double),
        ("ipddp_mu_kappa_epsilon", C.c_double),
        ("ipddp_check_state_stationarity", C.c_int32), ("ipddp_theta_norm_l2", C.c_int32),
        ("ipddp_max_filter_size", C.c_int32), ("ipddp_warmstart_repair", C.c_int32),
        ("ipddp_theta_0_floor", C.c_double), ("ipddp_warmstart_s_min", C.c_double),
        ("ipddp_warmstart_y_min", C.c_double), ("ipddp_warmstart_interior_factor", C.c_double),
        ("ipddp_jacobian_regularization_value", C.c_double),
        ("ipddp_jacobian_regularization_exponent", C.c_double),
        ("barrier_mu_initial", C.c_double), ("barrier_mu_min_value", C.c_double),
        ("barrier_mu_update_factor", C.c_double), ("barrier_mu_update_power", C.c_double),
        ("barrier_min_fraction_to_boundary", C.c_double),
        ("barrier_strategy", C.c_int32), ("_pad3", C.c_int32),
        ("max_cpu_time", C.c_double),
        ("logddp_mu_initial", C.c_double), ("logddp_mu_min_value", C.c_double), ("logddp_mu_update_factor", C.c_double),
        ("logddp_relaxed_delta", C.c_double),
        ("msipddp_costate_var_init_scale", C.c_double), ("msipddp_segment_length", C.c_int32), ("msipddp_rollout_type", C.c_int32),
        ("msipddp_use_controlled_rollout", C.c_int32), ("_pad4", C.c_int32),
    ]


def default_options():
    """Reference defaults: include/cddp-cpp/cddp_core/options.hpp:41-251, boxqp.hpp:30-41."""
    o = Options()
    o.tolerance = 1e-5; o.acceptable_tolerance = 1e-6; o.max_iterations = 1; o.use_ilqr = 1
    o.enable_parallel = 0; o.return_iteration_info = 0; o.warm_start = 0
    o.termination_scaling_max_factor = 100.0
    o.ls_max_iterations = 11; o.ls_initial_step_size = 1.0; o.ls_min_step_size = 1e-8
    o.ls_step_reduction_factor = 0.5
    o.reg_initial_value = 1e-6; o.reg_update_factor = 10.0; o.reg_max_value = 1e7; o.reg_min_value = 1e-10
    o.boxqp_max_iterations = 100; o.boxqp_min_gradient_norm = 1e-8
    o.boxqp_min_relative_improvement = 1e-8; o.boxqp_step_decrease_factor = 0.6
    o.boxqp_min_step_size = 1e-22; o.boxqp_armijo_constant = 0.1
    o.filter_merit_acceptance_threshold = 1e-6; o.filter_violation_acceptance_threshold = 1e-6
    o.filter_max_violation_threshold = 1e4; o.filter_min_violation_for_armijo_check = 1e-7
    o.filter_armijo_constant = 1e-4
    o.ipddp_dual_var_init_scale = 0.1; o.ipddp_slack_var_init_scale = 1e-2
    o.ipddp_barrier_tol_mult = 0.1; o.ipddp_barrier_update_dual_weight = 0.01
    o.ipddp_mu_kappa_epsilon = 10.0; o.ipddp_check_state_stationarity = 0; o.ipddp_theta_norm_l2 = 0
    o.ipddp_max_filter_size = 5; o.ipddp_warmstart_repair = 0; o.ipddp_theta_0_floor = 1.0
    o.ipddp_warmstart_s_min = 1e-4; o.ipddp_warmstart_y_min = 1e-4
    o.ipddp_warmstart_interior_factor = 1.1
    o.ipddp_jacobian_regularization_value = 1e-8; o.ipddp_jacobian_regularization_exponent = 0.25
    o.barrier_mu_initial = 1.0; o.barrier_mu_min_value = 1e-10; o.barrier_mu_update_factor = 0.5
    o.barrier_mu_update_power = 1.2; o.barrier_min_fraction_to_boundary = 0.99; o.barrier_strategy = 0
    o.max_cpu_time = 0.0
    o.logddp_mu_initial = 1.0; o.logddp_mu_min_value = 1e-10; o.logddp_mu_update_factor = 0.5; o.logddp_relaxed_delta = 1e-10
    o.msipddp_costate_var_init_scale = 1e-6; o.msipddp_segment_length = 5; o.msipddp_rollout_type = 0; o.msipddp_use_controlled_rollout = 0
    return o


class Constraint(C.Structure):
    _fields_ = [
        ("name", C.c_char * NAME_LEN), ("kind", C.c_int32), ("dim", C.c_int32),
        ("lower", _dp), ("upper", _dp), ("center", _dp), ("A", _dp), ("b", _dp),
        ("radius", C.c_double), ("scale", C.c_double),
    ]


class TerminalConstraint(C.Structure):
    _fields_ = [
        ("name", C.c_char * NAME_LEN), ("kind", C.c_int32), ("dim", C.c_int32),
        ("target", _dp), ("A", _dp), ("b", _dp),
    ]


class ProblemStruct(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("solver", C.c_int32), ("model", C.c_int32), ("integrator", C.c_int32),
        ("nx", C.c_int32), ("nu", C.c_int32), ("horizon", C.c_int32), ("_pad0", C.c_int32),
        ("dt", C.c_double), ("model_params", C.c_double * MAX_MODEL_PARAMS),
        ("lti_A", _dp), ("lti_B", _dp),
        ("Q", _dp), ("R", _dp), ("Qf", _dp), ("x_ref", _dp), ("x_ref_traj", _dp),
        ("n_constraints", C.c_int32), ("n_terminal", C.c_int32),
        ("constraints", C.POINTER(Constraint)), ("terminal", C.POINTER(TerminalConstraint)),
        ("options", Options),
    ]


class Result(C.Structure):
    _fields_ = [
        ("final_objective", C.c_double), ("merit_function", C.c_double),
        ("inf_pr", C.c_double), ("inf_du", C.c_double), ("inf_comp", C.c_double),
        ("barrier_mu", C.c_double), ("regularization", C.c_double),
        ("alpha_pr", C.c_double), ("alpha_du", C.c_double), ("step_norm", C.c_double),
        ("iterations", C.c_int32), ("status", C.c_int32), ("n_backward", C.c_int32), ("n_forward", C.c_int32),
    ]


RESULT_DTYPE = np.dtype([
    ("final_objective", "f8"), ("merit_function", "f8"), ("inf_pr", "f8"), ("inf_du", "f8"),
    ("inf_comp", "f8"), ("barrier_mu", "f8"), ("regularization", "f8"), ("alpha_pr", "f8"),
    ("alpha_du", "f8"), ("step_norm", "f8"), ("iterations", "i4"), ("status", "i4"),
    ("n_backward", "i4"), ("n_forward", "i4"),
])
TRIAL_DTYPE = np.dtype([
    ("alpha", "f8"), ("alpha_pr", "f8"), ("alpha_du", "f8"), ("cost", "f8"), ("merit_function", "f8"),
    ("theta", "f8"), ("inf_pr", "f8"), ("inf_comp", "f8"), ("success", "i4"), ("_pad", "i4"),
])
GATHER_DTYPE = np.dtype([("final_objective", "f8"), ("iterations", "i4"), ("status", "i4")])


TIMING_ROLLOUT, TIMING_ALL, TIMING_SWEEP = 0, 1, 2


class Stats(C.Structure):
    _fields_ = [
        ("solve_ms", C.c_double), ("backward_ms", C.c_double), ("forward_ms", C.c_double),
        ("update_ms", C.c_double), ("sweeps", C.c_int64), ("rollouts", C.c_int64),
        ("rollouts_launched", C.c_int64), ("traj_iterations", C.c_int64),
        ("outer_iterations", C.c_int32), ("n_converged", C.c_int32), ("kernel_launches", C.c_int32),
        ("timing_detail", C.c_int32), ("rollout_steps", C.c_int64),
    ]


def _arr(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class Problem:
    """Python-side owner of a cddp_hip_problem descriptor (keeps the numpy buffers alive)."""

    def __init__(self, solver, model, integrator, nx, nu, horizon, dt, Q, R, Qf, x_ref,
                 model_params=(), lti_A=None, lti_B=None, x_ref_traj=None, options=None):
        self.keep = []
        self.c = ProblemStruct()
        self.c.abi_version = ABI_VERSION
        self.c.solver = solver; self.c.model = model; self.c.integrator = integrator
        self.c.nx = nx; self.c.nu = nu; self.c.horizon = horizon; self.c.dt = dt
        for i, v in enumerate(model_params):
            self.c.model_params[i] = float(v)
        self.Q = _arr(Q).reshape(nx, nx); self.R = _arr(R).reshape(nu, nu); self.Qf = _arr(Qf).reshape(nx, nx)
        self.x_ref = _arr(x_ref).reshape(nx)
        self.c.Q = _ptr(self.Q); self.c.R = _ptr(self.R); self.c.Qf = _ptr(self.Qf); self.c.x_ref = _ptr(self.x_ref)
        if lti_A is not None:
            self.lti_A = _arr(lti_A).reshape(nx, nx); self.lti_B = _arr(lti_B).reshape(nx, nu)
            self.c.lti_A = _ptr(self.lti_A); self.c.lti_B = _ptr(self.lti_B)
        if x_ref_traj is not None:
            self.x_ref_traj = _arr(x_ref_traj).reshape(horizon + 1, nx)
            self.c.x_ref_traj = _ptr(self.x_ref_traj)
        self.c.options = options if options is not None else default_options()
        self._cons = []
        self._terms = []
        self.nx, self.nu, self.N, self.dt = nx, nu, horizon, dt

    @property
    def options(self):
        return self.c.options

    def _rebuild(self):
        if self._cons:
            arr = (Constraint * len(self._cons))(*self._cons)
            self._cons_arr = arr
            self.c.constraints = C.cast(arr, C.POINTER(Constraint)); self.c.n_constraints = len(self._cons)
        else:
            self.c.n_constraints = 0
        if self._terms:
            arr = (TerminalConstraint * len(self._terms))(*self._terms)
            self._terms_arr = arr
            self.c.terminal = C.cast(arr, C.POINTER(TerminalConstraint)); self.c.n_terminal = len(self._terms)
        else:
            self.c.n_terminal = 0

    def add_control_box(self, name, lower, upper, scale=1.0):
        lo, up = _arr(lower), _arr(upper); self.keep += [lo, up]
        c = Constraint(); c.name = name.encode(); c.kind = CON_CONTROL_BOX; c.dim = lo.size
        c.lower = _ptr(lo); c.upper = _ptr(up); c.scale = scale
        self._cons.append(c); self._rebuild(); return self

    def add_state_box(self, name, lower, upper, scale=1.0):
        lo, up = _arr(lower), _arr(upper); self.keep += [lo, up]
        c = Constraint(); c.name = name.encode(); c.kind = CON_STATE_BOX; c.dim = lo.size
        c.lower = _ptr(lo); c.upper = _ptr(up); c.scale = scale
        self._cons.append(c); self._rebuild(); return self

    def add_ball(self, name, radius, center, scale=1.0):
        ce = _arr(center); self.keep += [ce]
        c = Constraint(); c.name = name.encode(); c.kind = CON_BALL; c.dim = ce.size
        c.center = _ptr(ce); c.radius = radius; c.scale = scale
        self._cons.append(c); self._rebuild(); return self

    def add_linear(self, name, A, b):
        A_, b_ = _arr(A), _arr(b); self.keep += [A_, b_]
        c = Constraint(); c.name = name.encode(); c.kind = CON_LINEAR; c.dim = b_.size
        c.A = _ptr(A_); c.b = _ptr(b_); c.scale = 1.0
        self._cons.append(c); self._rebuild(); return self

    def add_second_order_cone(self, name, cone_origin, opening_direction, cone_angle_fov, regularization_epsilon=1e-6):
        """SecondOrderConeConstraint (constraint.hpp:626-668): the constructor's checks and normalisation happen here."""
        if cone_angle_fov < 0 or cone_angle_fov > np.pi:
            raise ValueError("SecondOrderConeConstraint: Cone angle must be between 0 and PI.")
        if regularization_epsilon <= 0:
            raise ValueError("SecondOrderConeConstraint: Regularization epsilon must be positive.")
        o = _arr(cone_origin).reshape(3); a = _arr(opening_direction).reshape(3)
        n = float(np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]))
        if n == 0.0:
            raise ValueError("SecondOrderConeConstraint: Opening direction cannot be zero vector.")
        a = a / n
        self.keep += [o, a]
        c = Constraint(); c.name = name.encode(); c.kind = CON_SOC; c.dim = 3
        c.center = _ptr(o); c.lower = _ptr(a); c.radius = float(np.cos(cone_angle_fov)); c.scale = regularization_epsilon
        self._cons.append(c); self._rebuild(); return self

    def add_thrust_magnitude(self, name, min_thrust_norm, max_thrust_norm, epsilon=1e-6):
        """ThrustMagnitudeConstraint (constraint.hpp:802-838)."""
        if min_thrust_norm < 0.0:
            raise ValueError("ThrustMagnitudeConstraint: min_thrust_norm must be non-negative.")
        if max_thrust_norm < min_thrust_norm:
            raise ValueError("ThrustMagnitudeConstraint: max_thrust_norm must be greater than or equal to min_thrust_norm.")
        if epsilon <= 0.0:
            raise ValueError("ThrustMagnitudeConstraint: epsilon must be positive.")
        mn = _arr([min_thrust_norm]); self.keep += [mn]
        c = Constraint(); c.name = name.encode(); c.kind = CON_THRUST; c.dim = self.nu
        c.lower = _ptr(mn); c.radius = max_thrust_norm; c.scale = epsilon
        self._cons.append(c); self._rebuild(); return self

    def add_max_thrust_magnitude(self, name, max_thrust_norm, epsilon=1e-6):
        """MaxThrustMagnitudeConstraint (constraint.hpp:929-953)."""
        if max_thrust_norm < 0.0:
            raise ValueError("MaxThrustMagnitudeConstraint: max_thrust_norm must be non-negative.")
        if epsilon <= 0.0:
            raise ValueError("MaxThrustMagnitudeConstraint: epsilon must be positive.")
        c = Constraint(); c.name = name.encode(); c.kind = CON_MAX_THRUST; c.dim = self.nu
        c.radius = max_thrust_norm; c.scale = epsilon
        self._cons.append(c); self._rebuild(); return self

    def add_terminal_equality(self, name, target):
        t_ = _arr(target); self.keep += [t_]
        c = TerminalConstraint(); c.name = name.encode(); c.kind = TERM_EQUALITY; c.dim = t_.size; c.target = _ptr(t_)
        self._terms.append(c); self._rebuild(); return self

    def add_terminal_inequality(self, name, A, b):
        A_, b_ = _arr(A), _arr(b); self.keep += [A_, b_]
        c = TerminalConstraint(); c.name = name.encode(); c.kind = TERM_INEQUALITY; c.dim = b_.size
        c.A = _ptr(A_); c.b = _ptr(b_)
        self._terms.append(c); self._rebuild(); return self

    def dual_dim(self):
        m = 0
        for c in self._cons:
            m += {CON_CONTROL_BOX: 2 * c.dim, CON_STATE_BOX: 2 * c.dim, CON_BALL: 1, CON_LINEAR: c.dim, CON_SOC: 1, CON_THRUST: 2, CON_MAX_THRUST: 1}[c.kind]
        return m


# ----------------------------------------------------------------------------------------------
# Problem builders for the BASELINE configs; constants from the
# reference examples (examples/cddp_pendulum.cpp:24-65, cddp_cartpole.cpp:24-66,
# cddp_unicycle.cpp / python_portfolio_lib.py:374-446, cddp_quadrotor_point.cpp:22-96,
# cddp_manipulator.cpp:22-68).
# ----------------------------------------------------------------------------------------------
def pendulum_problem(solver=SOLVER_IPDDP, constrained=True, horizon=100):
    o = default_options(); o.max_iterations = 30; o.tolerance = 1e-4; o.acceptable_tolerance = 1e-5
    o.reg_initial_value = 1e-6
    dt = 0.02
    p = Problem(solver, MODEL_PENDULUM, EULER, 2, 1, horizon, dt, np.zeros((2, 2)), 0.1 * np.eye(1),
                100.0 * np.eye(2), [0.0, 0.0], model_params=[0.5, 1.0, 0.01, 9.81], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", [-20.0], [20.0])
    p.x0 = np.array([np.pi, 0.0])
    return p


def cartpole_problem(solver=SOLVER_IPDDP, constrained=True, horizon=100):
    o = default_options(); o.max_iterations = 80; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-5
    o.reg_initial_value = 1e-5
    dt = 0.05
    p = Problem(solver, MODEL_CARTPOLE, RK4, 4, 1, horizon, dt, np.zeros((4, 4)), 0.1 * np.eye(1),
                100.0 * np.eye(4), [0.0, np.pi, 0.0, 0.0], model_params=[1.0, 0.2, 0.5, 9.81, 0.0], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", [-5.0], [5.0])
    p.x0 = np.zeros(4)
    return p


def unicycle_problem(solver=SOLVER_IPDDP, horizon=200, obstacle=True):
    o = default_options(); o.max_iterations = 100; o.tolerance = 1e-4; o.acceptable_tolerance = 1e-6
    dt = 0.03
    p = Problem(solver, MODEL_UNICYCLE, EULER, 3, 2, horizon, dt, np.zeros((3, 3)), 0.05 * np.eye(2),
                np.diag([100.0, 100.0, 50.0]), [2.0, 2.0, np.pi / 2], options=o)
    p.add_control_box("control_limits", [-1.1, -np.pi], [1.1, np.pi])
    if obstacle:
        p.add_ball("obstacle", 0.4, [1.0, 1.0])
    p.x0 = np.array([0.0, 0.0, np.pi / 4])
    p.U0_const = np.array([0.5, 0.1])
    return p


def hcw_problem(solver=SOLVER_IPDDP, horizon=80, constrained=True, integrator=None):
    """Hill-Clohessy-Wiltshire rendezvous (src/dynamics_model/spacecraft_linear.cpp; orbit of tests/dynamics_model/test_spacecraft_linear.cpp:
    500 km altitude, dt = 10 s, m = 1 kg as its DiscreteDynamics case): from its drifting initial state to the origin, thrust box."""
    o = default_options(); o.max_iterations = 40; o.tolerance = 1e-5; o.acceptable_tolerance = 1e-6; o.reg_initial_value = 1e-6
    n = float(np.sqrt(3.986004418e14 / (6371e3 + 500e3) ** 3))
    dt = 10.0
    p = Problem(solver, MODEL_HCW, RK4 if integrator is None else integrator, 6, 3, horizon, dt, np.diag([1e-4] * 3 + [1e-2] * 3), 1.0 * np.eye(3),
                np.diag([10.0] * 3 + [100.0] * 3), np.zeros(6), model_params=[n, 1.0], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", -0.5 * np.ones(3), 0.5 * np.ones(3))
    p.x0 = np.array([-37.59664132226163, 27.312455860666148, 13.656227930333074, 0.015161970413423813, 0.08348413138390476, 0.04174206569195238])
    return p


def bicycle_problem(solver=SOLVER_IPDDP, horizon=100, constrained=True, integrator=None):
    """Kinematic bicycle (src/dynamics_model/bicycle.cpp; constants of tests/dynamics_model/test_bicycle.cpp:28-40: dt 0.05, wheelbase
    2): drive from the origin to (2, 1, pi/4) at rest; state [x, y, theta, v], control [a, delta]."""
    o = default_options(); o.max_iterations = 60; o.tolerance = 1e-4; o.acceptable_tolerance = 1e-6
    dt = 0.05
    p = Problem(solver, MODEL_BICYCLE, EULER if integrator is None else integrator, 4, 2, horizon, dt, np.zeros((4, 4)), np.diag([0.05, 0.5]),
                np.diag([100.0, 100.0, 50.0, 10.0]), [2.0, 1.0, np.pi / 4, 0.0], model_params=[2.0], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", [-2.0, -0.5], [2.0, 0.5])
    p.x0 = np.array([0.0, 0.0, 0.0, 0.5])
    p.U0_const = np.array([0.1, 0.05])
    return p


def car_problem(solver=SOLVER_IPDDP, horizon=100, constrained=True):
    """The reference's car (src/dynamics_model/car.cpp, a DISCRETE plant; dt 0.03, wheelbase 2, control box +-[0.5, 2] of
    tests/cddp_core/test_ipddp_solver.cpp:686-750) with a QUADRATIC parking cost, so that the device-resident solve can be compared with
    the oracle (the reference's own test pairs it with a user NonlinearObjective: that shape runs through the plug-in solve)."""
    o = default_options(); o.max_iterations = 80; o.tolerance = 1e-4; o.acceptable_tolerance = 1e-6
    o.reg_initial_value = 1e-2
    dt = 0.03
    p = Problem(solver, MODEL_CAR, EULER, 4, 2, horizon, dt, np.diag([1e-2, 1e-2, 0.0, 0.0]), np.diag([1e-2, 1e-4]),
                np.diag([10.0, 10.0, 10.0, 3.0]), [0.0, 0.0, 0.0, 0.0], model_params=[2.0], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", [-0.5, -2.0], [0.5, 2.0])
    p.x0 = np.array([1.0, 1.0, 1.5 * np.pi, 0.0])
    p.U0_const = np.array([0.01, 0.1])
    return p


def unicycle_cone_problem(solver=SOLVER_IPDDP, horizon=100):
    """Unicycle with a control box and a SecondOrderConeConstraint on (x, y, theta) (constraint.hpp:626-800; geometry of
    tests/cddp_core/test_constraint.cpp:236-243: a 45-degree cone opening along +y from below the start), m = 5."""
    p = unicycle_problem(solver, horizon, obstacle=False)
    p._cons[0].name = b"ControlConstraint"; p._rebuild()   # std::map order: "ControlConstraint" < "SecondOrderConeConstraint" (the device layout); 'S' < 'c'
    p.add_second_order_cone("SecondOrderConeConstraint", [0.0, -0.5, 0.0], [0.0, 1.0, 0.0], np.pi / 4.0 + 0.35, 1e-6)
    return p


def unicycle_thrust_problem(solver=SOLVER_IPDDP, horizon=100, two_sided=True):
    """Unicycle whose control norm |(v, omega)| is bounded by the thrust-magnitude rows (constraint.hpp:802-1048): the two-sided
    ThrustMagnitudeConstraint (m = 2) or MaxThrustMagnitudeConstraint (m = 1) under the reference names.  Weights chosen so that the
    reference algorithm converges: IPDDP linearises path constraints to first order only (no constraint Hessians in
    ipddp_solver.cpp), and with a norm bound that ends up ACTIVE its iterates do not settle (oracle and twin agree on that)."""
    o = default_options(); o.max_iterations = 100; o.tolerance = 1e-4; o.acceptable_tolerance = 1e-6
    dt = 0.03
    p = Problem(solver, MODEL_UNICYCLE, EULER, 3, 2, horizon, dt, np.zeros((3, 3)), (0.5 if two_sided else 2.0) * np.eye(2),
                np.diag([10.0, 10.0, 5.0]), [2.0, 2.0, np.pi / 2], options=o)
    if two_sided:
        p.add_thrust_magnitude("ThrustMagnitudeConstraint", 0.3, 2.0, 1e-6)
    else:
        p.add_max_thrust_magnitude("MaxThrustMagnitudeConstraint", 2.0, 1e-6)
    p.x0 = np.array([0.0, 0.0, np.pi / 4])
    p.U0_const = np.array([0.5, 0.1])
    return p


def scalar_integrator_problem(horizon=8, path_constraint=False, terminal_inequality=False,
                              terminal_equality=False, options=None, solver=SOLVER_IPDDP):
    """LTI A=B=1 problems of tests/cddp_core/test_ipddp_solver.cpp:137-242, 1147-1637."""
    o = options if options is not None else default_options()
    p = Problem(solver, MODEL_LTI, EULER, 1, 1, horizon, 1.0, np.zeros((1, 1)), 1e-2 * np.eye(1),
                100.0 * np.eye(1), [1.0], lti_A=np.eye(1), lti_B=np.eye(1), options=o)
    if path_constraint:
        p.add_control_box("ControlConstraint", [-0.5], [0.5])
    if terminal_inequality:
        p.add_terminal_inequality("TerminalUpperBound", np.eye(1), np.zeros(1))
    if terminal_equality:
        p.add_terminal_equality("TerminalEquality", [0.0])
    p.x0 = np.zeros(1)
    return p


def quadrotor_problem(solver=SOLVER_IPDDP, horizon=120, constrained=True):
    o = default_options(); o.max_iterations = 120; o.ls_max_iterations = 15; o.reg_initial_value = 1e-4
    dt = 0.02
    Q = np.zeros((13, 13)); Q[4, 4] = Q[5, 5] = Q[6, 6] = 0.1
    R = 0.1 * np.eye(4)
    Qf = np.zeros((13, 13))
    for i in range(3): Qf[i, i] = 500.0
    for i in range(3, 7): Qf[i, i] = 1.0
    for i in range(7, 10): Qf[i, i] = 10.0
    goal = np.zeros(13); goal[0] = 3.0; goal[2] = 2.0; goal[3] = 1.0
    p = Problem(solver, MODEL_QUADROTOR, RK4, 13, 4, horizon, dt, Q, R, Qf, goal,
                model_params=[1.0, 0.2, 0.01, 0.01, 0.02, 9.81], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", np.zeros(4), 5.0 * np.ones(4))
    x0 = np.zeros(13); x0[3] = 1.0
    p.x0 = x0
    p.U0_const = (1.0 * 9.81 / 4.0) * np.ones(4)
    return p


def quadrotor_figure8_problem(solver=SOLVER_IPDDP, horizon=400):
    """The reference's own N = 400 quadrotor case: figure-8 tracking with per-step reference states
    (tests/cddp_core/test_ipddp_solver.cpp:887-1080, test_clddp_solver.cpp:570-763): m = 1.2, arm = 0.165, u in [0, 4]^4,
    Q = Qf = diag(1 x 7, 0 x 6), R = 0.01 I, hover U0, X0 = hover rollout.  It exercises the time-varying-reference branch of
    QuadraticObjective (objective.cpp:83-88)."""
    o = default_options(); o.tolerance = 1e-6; o.reg_initial_value = 1e-4; o.return_iteration_info = 1
    if solver == SOLVER_IPDDP:
        o.max_iterations = 500; o.acceptable_tolerance = 1e-5
    else:
        o.max_iterations = 200; o.acceptable_tolerance = 1e-6
    dt = 0.02
    Q = np.zeros((13, 13))
    for i in range(7): Q[i, i] = 1.0
    R = 0.01 * np.eye(4); Qf = Q.copy()
    omega = 2.0 * np.pi / (horizon * dt)
    ref = np.zeros((horizon + 1, 13))
    for i in range(horizon + 1):
        a = omega * (i * dt)
        ref[i, 0] = 3.0 * np.cos(a); ref[i, 1] = 3.0 * np.sin(a) * np.cos(a); ref[i, 2] = 2.0; ref[i, 3] = 1.0
    goal = np.zeros(13); goal[0] = 3.0; goal[2] = 2.0; goal[3] = 1.0
    p = Problem(solver, MODEL_QUADROTOR, RK4, 13, 4, horizon, dt, Q, R, Qf, goal,
                model_params=[1.2, 0.165, 7.782e-3, 7.782e-3, 1.439e-2, 9.81], x_ref_traj=ref, options=o)
    p.add_control_box("ControlConstraint", np.zeros(4), 4.0 * np.ones(4))
    p.x0 = goal.copy()
    p.U0_const = (1.2 * 9.81 / 4.0) * np.ones(4)
    return p


def quadrotor12_problem(solver=SOLVER_IPDDP, horizon=400, constrained=True):
    """SYNTHETIC throughput shape of BASELINE config 4 (nx=12, nu=4, N=400)."""
    o = default_options(); o.max_iterations = 120; o.ls_max_iterations = 15; o.reg_initial_value = 1e-4
    dt = 0.02
    Q = np.zeros((12, 12)); Q[6, 6] = Q[7, 7] = Q[8, 8] = 0.1
    R = 0.1 * np.eye(4)
    Qf = np.diag([500.0] * 3 + [10.0] * 3 + [1.0] * 3 + [0.0] * 3)
    goal = np.zeros(12); goal[0] = 3.0; goal[2] = 2.0
    p = Problem(solver, MODEL_QUADROTOR_EULER12, RK4, 12, 4, horizon, dt, Q, R, Qf, goal,
                model_params=[1.0, 0.2, 0.01, 0.01, 0.02, 9.81], options=o)
    if constrained:
        p.add_control_box("ControlConstraint", np.zeros(4), 5.0 * np.ones(4))
    p.x0 = np.zeros(12)
    p.U0_const = (1.0 * 9.81 / 4.0) * np.ones(4)
    return p


def manipulator_problem(solver=SOLVER_IPDDP, horizon=160, terminal_equality=False, constrained=True):
    """examples/cddp_manipulator.cpp:22-68 (3-DOF, rk4, FD Jacobians; X0 = linear interpolation)."""
    o = default_options(); o.max_iterations = 80; o.ls_max_iterations = 20
    dt = 0.01
    goal = np.array([np.pi, -np.pi / 6, -np.pi / 3, 0.0, 0.0, 0.0])
    Q = np.diag([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]); R = 0.1 * np.eye(3); Qf = 100.0 * Q
    p = Problem(solver, MODEL_MANIPULATOR, RK4, 6, 3, horizon, dt, Q, R, Qf, goal, options=o)
    if constrained:
        p.add_control_box("ControlConstraint", -50.0 * np.ones(3), 50.0 * np.ones(3))
    if terminal_equality:
        p.add_terminal_equality("TerminalEquality", goal)
    p.x0 = np.array([0.0, -np.pi / 2, np.pi, 0.0, 0.0, 0.0])
    al = np.linspace(0.0, 1.0, horizon + 1)[:, None]
    p.X0_single = (1.0 - al) * p.x0[None, :] + al * goal[None, :]
    return p


def manipulator7_problem(solver=SOLVER_IPDDP, horizon=150, terminal_equality=True, n_alphas=16):
    """SYNTHETIC throughput shape of BASELINE config 5 (nx=14, nu=7, N=150)."""
    o = default_options(); o.max_iterations = 80; o.tolerance = 1e-5; o.acceptable_tolerance = 1e-5
    o.ls_max_iterations = n_alphas; o.enable_parallel = 1
    dt = 0.01
    goal = np.concatenate([np.array([0.6, -0.4, 0.5, -0.3, 0.4, -0.2, 0.3]), np.zeros(7)])
    Q = np.zeros((14, 14)); R = 0.01 * np.eye(7); Qf = 100.0 * np.eye(14)
    p = Problem(solver, MODEL_MANIPULATOR7, RK4, 14, 7, horizon, dt, Q, R, Qf, goal, options=o)
    p.add_control_box("ControlConstraint", -50.0 * np.ones(7), 50.0 * np.ones(7))
    if terminal_equality:
        p.add_terminal_equality("TerminalEquality", goal)
    p.x0 = np.zeros(14)
    return p


def batch_x0(problem, batch, seed, spread=None):
    """Seeded per-trajectory x0 perturbations; trajectory 0 is the example."""
    rng = np.random.default_rng(seed)
    x0 = np.tile(problem.x0, (batch, 1)).astype(np.float64)
    if spread is None:
        spread = 0.1 * np.ones(problem.nx)
    pert = rng.uniform(-1.0, 1.0, size=(batch, problem.nx)) * np.asarray(spread)[None, :]
    pert[0] = 0.0
    return np.ascontiguousarray(x0 + pert)


def batch_U0(problem, batch):
    if hasattr(problem, "U0_const"):
        return np.ascontiguousarray(np.tile(problem.U0_const, (batch, problem.N, 1)).astype(np.float64))
    return None


# ----------------------------------------------------------------------------------------------
# Product binding
# ----------------------------------------------------------------------------------------------
_hip_libs = {}
# ONE library since round 4 (csrc/Makefile: every translation unit with -DCDDP_TRIG_SHARED): the plants' sin / cos / asin / tan and
# the solver core's log / pow are the straight-line routines of dev_trig.hpp -- the routines the CPU checker can run too (its
# trig_mode 1), which makes accept / reject decisions bit-comparable.  The `trig` arguments below are kept for callers written
# against rounds 1-3 (two libraries): None and "shared" name the library; "libm" is refused unless CDDP_HIP_LIB points at a
# hand-made device-libm experiment build (cddp_hip_trig_shared() == 0).


def default_trig():
    # (CDDP_HIP_LIB + CDDP_HIP_TRIG=libm: the hand-made device-libm comparison build of profiles/r04_trig_ab.md)
    return "libm" if (os.environ.get("CDDP_HIP_LIB") and os.environ.get("CDDP_HIP_TRIG") == "libm") else "shared"


def load_hip(trig=None):
    """Load the HIP C-ABI library.  Raises if it is missing: there is no fallback path."""
    trig = trig or default_trig()
    if trig not in ("shared", "libm"):
        raise ValueError("trig must be None, 'shared' or 'libm'")
    if "lib" not in _hip_libs:
        path = HIP_LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError("HIP library missing: %s -- run __graft_entry__.build(); "
                               "the product has no CPU fallback" % path)
        lib = C.CDLL(path)
        lib.cddp_hip_last_error.restype = C.c_char_p
        lib.cddp_hip_status_string.restype = C.c_char_p
        lib.cddp_hip_create.argtypes = [C.POINTER(ProblemStruct), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _hip_libs["lib"] = lib
    lib = _hip_libs["lib"]
    if (trig == "shared") != bool(lib.cddp_hip_trig_shared()):
        raise RuntimeError("trig=%r asked for, but %s was built %s -DCDDP_TRIG_SHARED" %
                           (trig, HIP_LIB_PATH, "with" if lib.cddp_hip_trig_shared() else "without"))
    return lib


EXPORTED_SYMBOLS = [
    "cddp_hip_default_options", "cddp_hip_abi_version", "cddp_hip_trig_shared", "cddp_hip_last_error", "cddp_hip_device_count",
    "cddp_hip_status_string", "cddp_hip_build_alphas", "cddp_hip_create", "cddp_hip_destroy",
    "cddp_hip_set_stream", "cddp_hip_set_initial", "cddp_hip_initialize", "cddp_hip_backward",
    "cddp_hip_forward", "cddp_hip_solve", "cddp_hip_get_results", "cddp_hip_get_trajectory",
    "cddp_hip_get_gains", "cddp_hip_get_value", "cddp_hip_get_linearization", "cddp_hip_get_duals", "cddp_hip_get_costates", "cddp_hip_get_backward_scalars",
    "cddp_hip_get_history", "cddp_hip_get_terminal", "cddp_hip_write_gather_records_device", "cddp_hip_dual_dim", "cddp_hip_batch",
    "cddp_hip_set_timing_detail", "cddp_hip_history_capacity", "cddp_hip_set_barrier_state", "cddp_hip_num_groups", "cddp_hip_concurrency", "cddp_hip_comm_unique_id", "cddp_hip_comm_init", "cddp_hip_comm_destroy", "cddp_hip_comm_info", "cddp_hip_get_plan_head", "cddp_hip_allgather_results",
    "cddp_hip_backward_stacks", "cddp_hip_stacks_create_abi", "cddp_hip_stacks_destroy", "cddp_hip_set_stacks", "cddp_hip_set_defect_stack", "cddp_hip_set_control_box", "cddp_hip_set_hessian_stacks", "cddp_hip_set_constraint_stacks",
    "cddp_hip_stacks_backward", "cddp_hip_stacks_last_kernel_ms", "cddp_hip_stacks_last_sweep_form", "cddp_hip_stacks_factor_cache", "cddp_hip_stacks_get_gains", "cddp_hip_stacks_get_constraint_gains",
    "cddp_hip_stacks_get_scalars", "cddp_hip_set_terminal_equality", "cddp_hip_stacks_get_terminal", "cddp_hip_plugin_solve", "cddp_hip_plugin_solve_terminal", "cddp_hip_plugin_set_host_threads", "cddp_hip_plugin_last_stats", "cddp_hip_model_eval", "cddp_hip_set_options", "cddp_hip_set_initial_state", "cddp_hip_forget_solver_state", "cddp_hip_set_duals", "cddp_hip_set_terminal",
]


class HipError(RuntimeError):
    pass


class HipBatchSolver:
    """Batch of independent trajectories of one problem on one GPU (C-ABI handle)."""

    def __init__(self, problem, batch, device=0, trig=None):
        self.lib = load_hip(trig)
        self.p = problem; self.B = batch
        self.h = C.c_void_p()
        rc = self.lib.cddp_hip_create(C.byref(problem.c), batch, device, C.byref(self.h))
        self._check(rc)
        self.m = self.lib.cddp_hip_dual_dim(self.h)

    def _check(self, rc):
        if rc != 0:
            raise HipError("cddp_hip error %d: %s" % (rc, self.lib.cddp_hip_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.cddp_hip_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_groups(self):
        return int(self.lib.cddp_hip_num_groups(self.h))

    def concurrency(self):
        """Groups cddp_hip_solve keeps in flight at once (2 = the static two-way CU partition of round 5)."""
        return int(self.lib.cddp_hip_concurrency(self.h))

    def set_stream(self, stream_ptr):
        self._check(self.lib.cddp_hip_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_timing_detail(self, detail):
        """TIMING_ROLLOUT (default): solve() brackets the rollout launches only; TIMING_ALL: every class;
        TIMING_SWEEP: derivative fill + sweep only (each event costs ~5 us of queue time)."""
        self._check(self.lib.cddp_hip_set_timing_detail(self.h, int(detail)))

    def set_initial(self, x0, U0=None, X0=None):
        x0 = _arr(x0); assert x0.shape == (self.B, self.p.nx)
        U0 = _arr(U0) if U0 is not None else None; X0 = _arr(X0) if X0 is not None else None
        self._check(self.lib.cddp_hip_set_initial(self.h, _ptr(x0), _ptr(U0), _ptr(X0)))

    def initialize(self):
        self._check(self.lib.cddp_hip_initialize(self.h))

    # ---- warm start / MPC restarts
    def set_options(self, options):
        self._check(self.lib.cddp_hip_set_options(self.h, C.byref(options)))

    def set_warm_start(self, flag=True):
        self.p.options.warm_start = 1 if flag else 0
        self.set_options(self.p.options)

    def set_initial_state(self, x0):
        x0 = _arr(x0).reshape(self.B, self.p.nx)
        self._check(self.lib.cddp_hip_set_initial_state(self.h, _ptr(x0)))

    def forget_solver_state(self):
        """As if the reference's solver object were new: the next warm-started initialize takes the "provided trajectory" branch."""
        self._check(self.lib.cddp_hip_forget_solver_state(self.h))

    def set_duals(self, S=None, Y=None):
        S = _arr(S) if S is not None else None; Y = _arr(Y) if Y is not None else None
        self._check(self.lib.cddp_hip_set_duals(self.h, _ptr(S), _ptr(Y)))

    def set_barrier_state(self, mu=None, reg=None):
        mu = _arr(mu).reshape(self.B) if mu is not None else None
        reg = _arr(reg).reshape(self.B) if reg is not None else None
        self._check(self.lib.cddp_hip_set_barrier_state(self.h, _ptr(mu), _ptr(reg)))

    def set_terminal_state(self, S_T=None, Y_T=None, Lambda_T=None):
        a = [(_arr(v) if v is not None else None) for v in (S_T, Y_T, Lambda_T)]
        self._check(self.lib.cddp_hip_set_terminal(self.h, _ptr(a[0]), _ptr(a[1]), _ptr(a[2])))

    def backward(self):
        ok = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.cddp_hip_backward(self.h, ok.ctypes.data_as(C.POINTER(C.c_int32))))
        return ok

    def forward(self, alphas):
        a = _arr(alphas); t = np.zeros((self.B, a.size), dtype=TRIAL_DTYPE)
        self._check(self.lib.cddp_hip_forward(self.h, _ptr(a), a.size, t.ctypes.data_as(C.c_void_p)))
        return t

    def solve(self):
        st = Stats()
        self._check(self.lib.cddp_hip_solve(self.h, C.byref(st)))
        return st

    def results(self):
        r = np.zeros(self.B, dtype=RESULT_DTYPE)
        self._check(self.lib.cddp_hip_get_results(self.h, r.ctypes.data_as(C.c_void_p)))
        return r

    def trajectory(self):
        X = np.zeros((self.B, self.p.N + 1, self.p.nx)); U = np.zeros((self.B, self.p.N, self.p.nu))
        self._check(self.lib.cddp_hip_get_trajectory(self.h, _ptr(X), _ptr(U)))
        return X, U

    def plan_head(self):
        """(u_0, x_1) of every trajectory's current iterate: what an MPC loop reads back (cddp_hip_get_plan_head)."""
        u0 = np.zeros((self.B, self.p.nu)); x1 = np.zeros((self.B, self.p.nx))
        self._check(self.lib.cddp_hip_get_plan_head(self.h, _ptr(u0), _ptr(x1)))
        return u0, x1

    def gains(self):
        K = np.zeros((self.B, self.p.N, self.p.nu, self.p.nx)); k = np.zeros((self.B, self.p.N, self.p.nu))
        self._check(self.lib.cddp_hip_get_gains(self.h, _ptr(K), _ptr(k)))
        return K, k

    def value(self):
        Vx = np.zeros((self.B, self.p.N + 1, self.p.nx)); Vxx = np.zeros((self.B, self.p.N + 1, self.p.nx, self.p.nx))
        self._check(self.lib.cddp_hip_get_value(self.h, _ptr(Vx), _ptr(Vxx)))
        return Vx, Vxx

    def linearization(self):
        """A_t = I + dt f_x, B_t = dt f_u of the last backward pass (F_x_, F_u_ of cddp_solver_base.cpp:319-394)."""
        A = np.zeros((self.B, self.p.N, self.p.nx, self.p.nx)); Bm = np.zeros((self.B, self.p.N, self.p.nx, self.p.nu))
        self._check(self.lib.cddp_hip_get_linearization(self.h, _ptr(A), _ptr(Bm)))
        return A, Bm

    def duals(self):
        S = np.zeros((self.B, self.p.N, self.m)); Y = np.zeros_like(S); G = np.zeros_like(S)
        if self.m:
            self._check(self.lib.cddp_hip_get_duals(self.h, _ptr(S), _ptr(Y), _ptr(G)))
        return S, Y, G

    def costates(self):
        """Costate trajectory Lambda of the current iterate, (B, rows, nx): rows = N + 1 (IPDDP) or N (MSIPDDP) (cddp_hip_get_costates)."""
        rows = C.c_int32(0)
        self._check(self.lib.cddp_hip_get_costates(self.h, None, C.byref(rows)))
        L = np.zeros((self.B, rows.value, self.p.nx))
        self._check(self.lib.cddp_hip_get_costates(self.h, _ptr(L), C.byref(rows)))
        return L

    def terminal(self):
        dims = np.zeros(2, dtype=np.int32)
        self._check(self.lib.cddp_hip_get_terminal(self.h, None, None, None, None, dims.ctypes.data_as(C.POINTER(C.c_int32))))
        mT, pT = int(dims[0]), int(dims[1])
        S = np.zeros((self.B, mT)); Y = np.zeros((self.B, mT)); G = np.zeros((self.B, mT)); L = np.zeros((self.B, pT))
        self._check(self.lib.cddp_hip_get_terminal(self.h, _ptr(S), _ptr(Y), _ptr(G), _ptr(L), None))
        return S, Y, G, L

    def backward_scalars(self):
        dV = np.zeros((self.B, 2)); reg = np.zeros(self.B)
        self._check(self.lib.cddp_hip_get_backward_scalars(self.h, _ptr(dV), _ptr(reg)))
        return dV, reg

    def history(self, hist_batch=1):
        cap = self.lib.cddp_hip_history_capacity(self.h)   # fixed at create time (set_options may lower max_iterations)
        h = np.zeros((hist_batch, cap, 9)); cnt = np.zeros(hist_batch, dtype=np.int32)
        self._check(self.lib.cddp_hip_get_history(self.h, hist_batch, _ptr(h), cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return [h[b, :cnt[b]].copy() for b in range(hist_batch)]

    def allgather_results(self, comm, world, shard_capacity, recv_device_ptr):
        """The path's single collective through the C-ABI (RCCL all-gather of the 16-byte records); comm = a
        communicator from `comm_init` (or None for world == 1)."""
        self._check(self.lib.cddp_hip_allgather_results(self.h, C.c_void_p(comm), int(world), int(shard_capacity), C.c_void_p(recv_device_ptr)))

    def write_gather_records_device(self, device_ptr):
        self._check(self.lib.cddp_hip_write_gather_records_device(self.h, C.c_void_p(device_ptr)))


def model_eval(model, integrator, dt, params, nx, nu, x, u, want=("step",), trig=None):
    """cddp_hip_model_eval: host evaluation of a built-in plant at one (x, u).  want: any of "step" (x_next), "jac" (f_x, f_u
    continuous-time), "hess" (f_xx[nx][nx][nx], f_uu[nx][nu][nu], f_ux[nx][nu][nx]).  Returns a dict."""
    lib = load_hip(trig)
    pv = np.zeros(MAX_MODEL_PARAMS); pv[:len(params)] = params
    x = _arr(x).reshape(nx); u = _arr(u).reshape(nu)
    out = {}
    xn = np.empty(nx) if "step" in want else None
    fx = np.empty((nx, nx)) if "jac" in want else None; fu = np.empty((nx, nu)) if "jac" in want else None
    fxx = np.empty((nx, nx, nx)) if "hess" in want else None; fuu = np.empty((nx, nu, nu)) if "hess" in want else None
    fux = np.empty((nx, nu, nx)) if "hess" in want else None
    lib.cddp_hip_model_eval.restype = C.c_int
    rc = lib.cddp_hip_model_eval(int(model), int(integrator), C.c_double(dt), _ptr(pv), int(nx), int(nu), _ptr(x), _ptr(u),
                                 _ptr(xn), _ptr(fx), _ptr(fu), _ptr(fxx), _ptr(fuu), _ptr(fux))
    if rc != 0:
        lib.cddp_hip_last_error.restype = C.c_char_p
        raise HipError(lib.cddp_hip_last_error().decode())
    if xn is not None: out["step"] = xn
    if fx is not None: out["jac"] = (fx, fu)
    if fxx is not None: out["hess"] = (fxx, fuu, fux)
    return out


# ---- host plug-in solve (include/cddp_hip.h: cddp_hip_plugin / cddp_hip_plugin_solve) ----------------------------------------
PLUGIN_MAX_CONSTRAINTS = 8
_F_DYN = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_double, _dp)
_F_JAC = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_double, _dp, _dp)
_F_HES = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_double, _dp, _dp, _dp)
_F_RC = C.CFUNCTYPE(C.c_double, C.c_void_p, _dp, _dp, C.c_int)
_F_TC = C.CFUNCTYPE(C.c_double, C.c_void_p, _dp)
_F_RCD = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp)
_F_TCD = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, _dp)
_F_CON = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_int, _dp, _dp, _dp)
_F_CHS = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_int, _dp, _dp, _dp)


class PluginStruct(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("options_bytes", C.c_int32), ("abort_flag", C.POINTER(C.c_int32)),
        ("user", C.c_void_p), ("nx", C.c_int32), ("nu", C.c_int32), ("n_constraints", C.c_int32),
        ("constraint_dims", C.c_int32 * PLUGIN_MAX_CONSTRAINTS),
        ("discrete_dynamics", _F_DYN), ("jacobians", _F_JAC), ("hessians", _F_HES),
        ("running_cost", _F_RC), ("terminal_cost", _F_TC), ("running_cost_derivatives", _F_RCD),
        ("terminal_cost_derivatives", _F_TCD), ("constraints", _F_CON),
        ("control_lower", _dp), ("control_upper", _dp), ("constraint_hessians", _F_CHS),
    ]


_F_TERM = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, _dp)


class PluginTerminalStruct(C.Structure):   # cddp_hip_plugin_terminal
    _fields_ = [("n_terminal", C.c_int32), ("dims", C.c_int32 * PLUGIN_MAX_CONSTRAINTS), ("equality", C.c_int32 * PLUGIN_MAX_CONSTRAINTS), ("evaluate", _F_TERM)]


def plugin_solve(solver, nx, nu, horizon, dt, options, x0, U0=None, X0=None, *, discrete_dynamics, jacobians, running_cost, terminal_cost,
                 running_cost_derivatives, terminal_cost_derivatives, hessians=None, constraints=None, constraint_dims=(),
                 control_lower=None, control_upper=None, constraint_hessians=None, device=0, trig=None,
                 terminal=None, terminal_dims=(), terminal_equality=(), want_terminal=False):
    """CDDP::solve() for HOST plug-ins through the C-ABI (cddp_hip_plugin_solve): the callables are the reference's virtual functions
    on numpy vectors -- discrete_dynamics(x, u, t) -> x_next; jacobians(x, u, t) -> (f_x, f_u) continuous-time;
    hessians(x, u, t) -> (f_xx[nx][nx][nx], f_uu[nx][nu][nu], f_ux[nx][nu][nx]); running_cost(x, u, index) -> float;
    terminal_cost(x) -> float; running_cost_derivatives(x, u, index) -> (l_x, l_u, l_xx, l_uu, l_ux); terminal_cost_derivatives(x)
    -> (l_x, l_xx); constraints(x, u, index, want_jacobians) -> (g - upper, G_x, G_u) stacked in name order.  The GPU runs the
    batched backward passes, the host the forward passes.  An exception raised by a callable stops the solve and is re-raised here.
    solver = SOLVER_LOGDDP runs the reference's LogDDP on the same callbacks; constraint_hessians(x, u, index) -> (g_xx[m][nx][nx],
    g_uu[m][nu][nu], g_ux[m][nu][nx]) or None supplies constraint curvature to its relaxed log barrier.  Returns (results, X, U, K).
    terminal(x_N, want_jacobian) -> (r, r_x) with the residual rows of every terminal-constraint object stacked in name order (terminal_dims,
    terminal_equality: 1 = TerminalEqualityConstraint-like, 0 = inequality g_T <= 0) runs cddp_hip_plugin_solve_terminal; want_terminal adds a
    fifth return value, the dict {S_T, Y_T, Lambda_T} of the final iterate."""
    lib = load_hip(trig)
    x0 = _arr(x0).reshape(-1, nx); B = x0.shape[0]; N = int(horizon)
    U0 = _arr(U0).reshape(B, N, nu) if U0 is not None else None
    X0 = _arr(X0).reshape(B, N + 1, nx) if X0 is not None else None
    m = int(sum(constraint_dims))
    err = []

    def vec(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,))

    abort = C.c_int32(0)

    def guard(fn, default=None):
        def w(*a):
            if err:
                return default
            try:
                return fn(*a)
            except BaseException as e:   # noqa: B902 -- re-raised by the caller thread after the C call returns
                err.append(e)
                abort.value = 1      # cddp_hip_plugin::abort_flag: the C solve returns at its next check instead of iterating on stale data
                return default
        return w

    def _dyn(_, x, u, t, out):
        vec(out, nx)[:] = np.asarray(discrete_dynamics(vec(x, nx).copy(), vec(u, nu).copy(), t), dtype=np.float64).reshape(nx)

    def _jac(_, x, u, t, fx, fu):
        a, b = jacobians(vec(x, nx).copy(), vec(u, nu).copy(), t)
        vec(fx, nx * nx)[:] = np.asarray(a, dtype=np.float64).reshape(nx * nx); vec(fu, nx * nu)[:] = np.asarray(b, dtype=np.float64).reshape(nx * nu)

    def _hes(_, x, u, t, fxx, fuu, fux):
        a, b, c2 = hessians(vec(x, nx).copy(), vec(u, nu).copy(), t)
        vec(fxx, nx * nx * nx)[:] = np.asarray(a, dtype=np.float64).reshape(-1); vec(fuu, nx * nu * nu)[:] = np.asarray(b, dtype=np.float64).reshape(-1)
        vec(fux, nx * nu * nx)[:] = np.asarray(c2, dtype=np.float64).reshape(-1)

    def _rc(_, x, u, idx):
        return float(running_cost(vec(x, nx).copy(), vec(u, nu).copy(), idx))

    def _tc(_, x):
        return float(terminal_cost(vec(x, nx).copy()))

    def _rcd(_, x, u, idx, lx, lu, lxx, luu, lux):
        a = running_cost_derivatives(vec(x, nx).copy(), vec(u, nu).copy(), idx)
        for dst, src, n in ((lx, a[0], nx), (lu, a[1], nu), (lxx, a[2], nx * nx), (luu, a[3], nu * nu), (lux, a[4], nu * nx)):
            vec(dst, n)[:] = np.asarray(src, dtype=np.float64).reshape(n)

    def _tcd(_, x, lx, lxx):
        a, b = terminal_cost_derivatives(vec(x, nx).copy())
        vec(lx, nx)[:] = np.asarray(a, dtype=np.float64).reshape(nx); vec(lxx, nx * nx)[:] = np.asarray(b, dtype=np.float64).reshape(nx * nx)

    def _con(_, x, u, idx, g, gx, gu):
        want = bool(gx) or bool(gu)
        gv, Gx, Gu = constraints(vec(x, nx).copy(), vec(u, nu).copy(), idx, want)
        vec(g, m)[:] = np.asarray(gv, dtype=np.float64).reshape(m)
        if gx: vec(gx, m * nx)[:] = np.asarray(Gx, dtype=np.float64).reshape(m * nx)
        if gu: vec(gu, m * nu)[:] = np.asarray(Gu, dtype=np.float64).reshape(m * nu)

    def _chs(_, x, u, idx, gxx, guu, gux):
        r = constraint_hessians(vec(x, nx).copy(), vec(u, nu).copy(), idx)
        if r is None:
            return
        vec(gxx, m * nx * nx)[:] = np.asarray(r[0], dtype=np.float64).reshape(-1); vec(guu, m * nu * nu)[:] = np.asarray(r[1], dtype=np.float64).reshape(-1)
        vec(gux, m * nu * nx)[:] = np.asarray(r[2], dtype=np.float64).reshape(-1)

    ps = PluginStruct()
    ps.nx, ps.nu, ps.n_constraints = nx, nu, len(constraint_dims)
    for i, dmy in enumerate(constraint_dims):
        ps.constraint_dims[i] = int(dmy)
    keep = [_F_DYN(guard(_dyn)), _F_JAC(guard(_jac)), _F_RC(guard(_rc, 0.0)), _F_TC(guard(_tc, 0.0)), _F_RCD(guard(_rcd)), _F_TCD(guard(_tcd))]
    ps.discrete_dynamics, ps.jacobians, ps.running_cost, ps.terminal_cost, ps.running_cost_derivatives, ps.terminal_cost_derivatives = keep
    if hessians is not None:
        keep.append(_F_HES(guard(_hes))); ps.hessians = keep[-1]
    if constraints is not None and m > 0:
        keep.append(_F_CON(guard(_con))); ps.constraints = keep[-1]
    if constraint_hessians is not None and m > 0:
        keep.append(_F_CHS(guard(_chs))); ps.constraint_hessians = keep[-1]
    lo = _arr(control_lower) if control_lower is not None else None
    up = _arr(control_upper) if control_upper is not None else None
    ps.control_lower, ps.control_upper = _ptr(lo), _ptr(up)
    ps.abi_version = ABI_VERSION; ps.options_bytes = C.sizeof(Options); ps.abort_flag = C.pointer(abort)
    res = np.zeros(B, dtype=RESULT_DTYPE)
    X = np.zeros((B, N + 1, nx)); U = np.zeros((B, N, nu)); K = np.zeros((B, N, nu, nx))
    tout = None
    if terminal is not None and len(terminal_dims) > 0:
        rows = int(sum(terminal_dims))
        mT = int(sum(d for d, e in zip(terminal_dims, terminal_equality) if not e)); pT = rows - mT

        def _term(_, x, r, rx):
            rv, Rx = terminal(vec(x, nx).copy(), bool(rx))
            vec(r, rows)[:] = np.asarray(rv, dtype=np.float64).reshape(rows)
            if rx: vec(rx, rows * nx)[:] = np.asarray(Rx, dtype=np.float64).reshape(rows * nx)

        ts = PluginTerminalStruct()
        ts.n_terminal = len(terminal_dims)
        for i, (dmy, e) in enumerate(zip(terminal_dims, terminal_equality)):
            ts.dims[i] = int(dmy); ts.equality[i] = 1 if e else 0
        keep.append(_F_TERM(guard(_term))); ts.evaluate = keep[-1]
        tbuf = np.zeros((B, max(1, 2 * mT + pT)))
        rc = lib.cddp_hip_plugin_solve_terminal(C.byref(ps), C.byref(ts), int(solver), N, C.c_double(dt), C.byref(options), int(device), B, _ptr(x0), _ptr(U0), _ptr(X0),
                                                res.ctypes.data_as(C.c_void_p), _ptr(X), _ptr(U), _ptr(K), _ptr(tbuf))
        tout = {"S_T": tbuf[:, :mT].copy(), "Y_T": tbuf[:, mT:2 * mT].copy(), "Lambda_T": tbuf[:, 2 * mT:2 * mT + pT].copy()}
    else:
        rc = lib.cddp_hip_plugin_solve(C.byref(ps), int(solver), N, C.c_double(dt), C.byref(options), int(device), B, _ptr(x0), _ptr(U0), _ptr(X0),
                                       res.ctypes.data_as(C.c_void_p), _ptr(X), _ptr(U), _ptr(K))
    if err:
        raise err[0]
    if rc != 0:
        raise HipError("cddp_hip error %d: %s" % (rc, lib.cddp_hip_last_error().decode()))
    if want_terminal:
        return res, X, U, K, tout
    return res, X, U, K


STACKS_CLDDP, STACKS_IPDDP, STACKS_IPDDP_PATH, STACKS_LOGDDP, STACKS_MSIPDDP, STACKS_MSIPDDP_PATH, STACKS_IPDDP_TERM_EQ = 0, 1, 2, 3, 4, 5, 6


class HipStackSolver:
    """Stack-fed backward pass bound to a handle (include/cddp_hip.h, "stack-fed mode"): the caller's host plugins fill the
    (N x batch) derivative stacks, the GPU sweeps them.  Arrays are batch-major numpy: fx (B,N,nx,nx), fu (B,N,nx,nu), lx (B,N,nx),
    lu (B,N,nu), lxx (B,N,nx,nx), luu (B,N,nu,nu), lux (B,N,nu,nx), VxN (B,nx), VxxN (B,nx,nx); y, s, g (B,N,m), Gx (B,N,m,nx),
    Gu (B,N,m,nu)."""

    def __init__(self, batch, nx, nu, m, horizon, device=0):
        self.lib = load_hip()
        self.B, self.nx, self.nu, self.m, self.N = batch, nx, nu, m, horizon
        self.h = C.c_void_p()
        self.lib.cddp_hip_stacks_last_kernel_ms.restype = C.c_double
        self._check(self.lib.cddp_hip_stacks_create_abi(ABI_VERSION, C.sizeof(Options), device, batch, nx, nu, m, horizon, C.byref(self.h)))

    def _check(self, rc):
        if rc != 0:
            raise HipError("cddp_hip error %d: %s" % (rc, self.lib.cddp_hip_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.cddp_hip_stacks_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stacks(self, fx=None, fu=None, lx=None, lu=None, lxx=None, luu=None, lux=None, VxN=None, VxxN=None):
        a = [(_arr(v) if v is not None else None) for v in (fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN)]
        self._check(self.lib.cddp_hip_set_stacks(self.h, *[_ptr(v) for v in a]))

    def set_defect_stack(self, defects=None):
        """Multiple-shooting defects f(x_t, u_t) - x_{t+1}, [B][N][nx] (MSIPDDP branch); None drops them."""
        a = _arr(defects) if defects is not None else None
        self._check(self.lib.cddp_hip_set_defect_stack(self.h, _ptr(a)))

    def set_control_box(self, lower=None, upper=None, U=None):
        """CLDDP branch: control bounds (nu each) and the current controls [B][N][nu]; all None removes the box."""
        a = [(_arr(v) if v is not None else None) for v in (lower, upper, U)]
        self._check(self.lib.cddp_hip_set_control_box(self.h, *[_ptr(v) for v in a]))

    def set_hessian_stacks(self, Fxx=None, Fuu=None, Fux=None):
        """dt-scaled dynamics Hessian tensors [B][N][nx][...] (full DDP); all None returns to Gauss-Newton."""
        a = [(_arr(v) if v is not None else None) for v in (Fxx, Fuu, Fux)]
        self._check(self.lib.cddp_hip_set_hessian_stacks(self.h, *[_ptr(v) for v in a]))

    def factor_cache(self, enable=True):
        """MSIPDDP's per-step factor cache (msipddp_solver.cpp:1169-1185): enable=True allocates / CLEARS it (start of a solve)."""
        self._check(self.lib.cddp_hip_stacks_factor_cache(self.h, 1 if enable else 0))

    def set_constraint_stacks(self, y=None, s=None, g=None, Gx=None, Gu=None):
        a = [(_arr(v) if v is not None else None) for v in (y, s, g, Gx, Gu)]
        self._check(self.lib.cddp_hip_set_constraint_stacks(self.h, *[_ptr(v) for v in a]))

    def set_terminal_equality(self, H_T, b_T, lambda_prev, reg_floor):
        """Terminal-equality branch (STACKS_IPDDP_TERM_EQ): H_T (B,pT,nx), b_T = -h_T (B,pT), lambda_prev (B,pT), reg_floor (B,)."""
        H = _arr(H_T); pT = H.shape[1]
        a = [_arr(b_T).reshape(self.B, pT), _arr(lambda_prev).reshape(self.B, pT), _arr(np.broadcast_to(np.asarray(reg_floor, dtype=np.float64), (self.B,)))]
        self.pT = pT
        self._check(self.lib.cddp_hip_set_terminal_equality(self.h, pT, _ptr(H), _ptr(a[0]), _ptr(a[1]), _ptr(a[2])))

    def terminal(self):
        dlam = np.zeros((self.B, self.pT)); dX = np.zeros((self.B, self.N + 1, self.nx))
        self._check(self.lib.cddp_hip_stacks_get_terminal(self.h, _ptr(dlam), _ptr(dX)))
        return dlam, dX

    def backward(self, branch, options, reg, mu=None, retry=False):
        reg = _arr(np.broadcast_to(np.asarray(reg, dtype=np.float64), (self.B,)))
        mu = _arr(np.broadcast_to(np.asarray(mu, dtype=np.float64), (self.B,))) if mu is not None else None
        ok = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.cddp_hip_stacks_backward(self.h, int(branch), C.byref(options), _ptr(reg), _ptr(mu), 1 if retry else 0,
                                                      ok.ctypes.data_as(C.POINTER(C.c_int32))))
        return ok

    def kernel_ms(self):
        return float(self.lib.cddp_hip_stacks_last_kernel_ms(self.h))

    def sweep_form(self):
        """0 = one lane per trajectory, 1 = lane-cooperative (the default for nx > 8)."""
        return int(self.lib.cddp_hip_stacks_last_sweep_form(self.h))

    def gains(self):
        B, N, nx, nu = self.B, self.N, self.nx, self.nu
        K = np.zeros((B, N, nu, nx)); k = np.zeros((B, N, nu)); Vx = np.zeros((B, N + 1, nx)); Vxx = np.zeros((B, N + 1, nx, nx)); dV = np.zeros((B, 2))
        self._check(self.lib.cddp_hip_stacks_get_gains(self.h, _ptr(K), _ptr(k), _ptr(Vx), _ptr(Vxx), _ptr(dV)))
        return K, k, Vx, Vxx, dV

    def constraint_gains(self):
        B, N, nx, m = self.B, self.N, self.nx, self.m
        ky = np.zeros((B, N, m)); Ky = np.zeros((B, N, m, nx)); ks = np.zeros((B, N, m)); Ks = np.zeros((B, N, m, nx)); dX = np.zeros((B, N + 1, nx))
        self._check(self.lib.cddp_hip_stacks_get_constraint_gains(self.h, _ptr(ky), _ptr(Ky), _ptr(ks), _ptr(Ks), _ptr(dX)))
        return ky, Ky, ks, Ks, dX

    def scalars(self):
        names = ("reg", "inf_du", "inf_pr", "inf_comp", "step_norm", "alpha_pr_max", "alpha_du_max")
        a = [np.zeros(self.B) for _ in names]
        self._check(self.lib.cddp_hip_stacks_get_scalars(self.h, *[_ptr(v) for v in a]))
        return dict(zip(names, a))


def hip_backward_stacks(fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN, reg, reg_in_value, device=0):
    lib = load_hip()
    fx = _arr(fx); B, N, nx, _ = fx.shape; nu = _arr(fu).shape[3]
    args = [_arr(a) for a in (fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN)]
    K = np.zeros((B, N, nu, nx)); k = np.zeros((B, N, nu)); Vx = np.zeros((B, N + 1, nx)); Vxx = np.zeros((B, N + 1, nx, nx))
    dV = np.zeros((B, 2)); ok = np.zeros(B, dtype=np.int32); ms = C.c_double(0.0)
    rc = lib.cddp_hip_backward_stacks(device, B, nx, nu, N, *[_ptr(a) for a in args], C.c_double(reg), int(reg_in_value),
                                      _ptr(K), _ptr(k), _ptr(Vx), _ptr(Vxx), _ptr(dV),
                                      ok.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms))
    if rc != 0:
        raise HipError("cddp_hip error %d: %s" % (rc, lib.cddp_hip_last_error().decode()))
    return K, k, Vx, Vxx, dV, ok, ms.value


# ---- RCCL communicator helpers of the C-ABI (include/cddp_hip.h, "multi-GPU") -------------------------------------
COMM_ID_BYTES = 128


def comm_unique_id():
    lib = load_hip()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.cddp_hip_comm_unique_id(buf)
    if rc != 0:
        raise HipError("cddp_hip error %d: %s" % (rc, lib.cddp_hip_last_error().decode()))
    return buf.raw


def comm_init(unique_id, world, rank, device):
    lib = load_hip()
    comm = C.c_void_p()
    rc = lib.cddp_hip_comm_init(C.c_char_p(bytes(unique_id)), int(world), int(rank), int(device), C.byref(comm))
    if rc != 0:
        raise HipError("cddp_hip error %d: %s" % (rc, lib.cddp_hip_last_error().decode()))
    return comm.value


def comm_info(comm):
    """(ranks RCCL sees behind the communicator, this process's rank)"""
    lib = load_hip()
    n = C.c_int(); r = C.c_int()
    rc = lib.cddp_hip_comm_info(C.c_void_p(comm), C.byref(n), C.byref(r))
    if rc != 0:
        raise HipError("cddp_hip error %d: %s" % (rc, lib.cddp_hip_last_error().decode()))
    return n.value, r.value


def comm_destroy(comm):
    if comm:
        load_hip().cddp_hip_comm_destroy(C.c_void_p(comm))
