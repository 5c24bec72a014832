"""pycddp-compatible Python front end of the MI355X solver core.

Same class / method / attribute names as the reference's pybind module (`python/src/bind_options.cpp:24-126`,
`bind_dynamics.cpp:104-247`, `bind_solver.cpp:519-663`, `python/pycddp/__init__.py`), so the scripts and tests
written against `pycddp` run on the GPU core after `import pycddp_amd as pycddp` -- for the plants, objectives and
constraint kinds the device has kernels for.  On top of it:

    solver.solve_batch(x0s, solver_type)   ->  list of CDDPSolution, one per row of x0s (one device-resident batch)

Everything goes through the C-ABI (`include/cddp_hip.h`); there is no CPU fallback.  Python subclasses of `DynamicalSystem`,
`Objective` / `NonlinearObjective` and `Constraint` (the reference's trampolines, `bind_dynamics.cpp:31-103`, `bind_objective.cpp`,
`bind_constraints.cpp`) run through the host plug-in solve (`cddp_hip_plugin_solve`: batched backward passes on the GPU, forward
passes on the host, callbacks into Python) -- as do single MSIPDDP / LogDDP solves for every problem (`solve_batch` of LogDDP or
MSIPDDP on a built-in plant with nx <= 8 runs on the resident kernels, switches `logddp_route` / `msipddp_route`); anything the core does not implement raises
(un-instantiated layouts, path-constrained MSIPDDP outside nu = 1 / nx = nu).
"""
import enum
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _api():
    name = "cddp_cpp_amd_pyapi"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_HERE, "pyapi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ enums / options
class SolverType(enum.Enum):        # options.hpp / bind_options.cpp:31-35
    CLDDP = "CLDDP"
    LogDDP = "LogDDP"
    IPDDP = "IPDDP"
    MSIPDDP = "MSIPDDP"


class BarrierStrategy(enum.IntEnum):   # bind_options.cpp:26-29
    ADAPTIVE = 0
    MONOTONIC = 1
    IPOPT = 2


class BoxQPOptions:                 # boxqp.hpp:30-41
    def __init__(self):
        self.max_iterations = 100; self.min_gradient_norm = 1e-8; self.min_relative_improvement = 1e-8
        self.step_decrease_factor = 0.6; self.min_step_size = 1e-22; self.armijo_constant = 0.1; self.verbose = False


class LineSearchOptions:            # options.hpp
    def __init__(self):
        self.max_iterations = 11; self.initial_step_size = 1.0; self.min_step_size = 1e-8; self.step_reduction_factor = 0.5


class RegularizationOptions:
    def __init__(self):
        self.initial_value = 1e-6; self.update_factor = 10.0; self.max_value = 1e7; self.min_value = 1e-10
        self.step_initial_value = 1.0


class BarrierOptions:
    def __init__(self):
        self.mu_initial = 1.0; self.mu_min_value = 1e-10; self.mu_update_factor = 0.5; self.mu_update_power = 1.2
        self.min_fraction_to_boundary = 0.99; self.strategy = BarrierStrategy.ADAPTIVE


class FilterOptions:
    def __init__(self):
        self.merit_acceptance_threshold = 1e-6; self.violation_acceptance_threshold = 1e-6
        self.max_violation_threshold = 1e4; self.min_violation_for_armijo_check = 1e-7; self.armijo_constant = 1e-4


class IPDDPOptions:
    def __init__(self):
        self.dual_var_init_scale = 0.1; self.slack_var_init_scale = 1e-2; self.barrier = BarrierOptions()


class LogBarrierOptions:            # options.hpp:135-143 (LogDDP)
    def __init__(self):
        self.use_relaxed_log_barrier_penalty = False; self.relaxed_log_barrier_delta = 1e-10; self.barrier = BarrierOptions()


class MSIPDDPOptions(IPDDPOptions):
    def __init__(self):
        super().__init__()
        self.segment_length = 5; self.rollout_type = "nonlinear"; self.use_controlled_rollout = False
        self.costate_var_init_scale = 1e-6


class CDDPOptions:                  # options.hpp:41-251 / bind_options.cpp:96-126
    def __init__(self):
        self.tolerance = 1e-5; self.acceptable_tolerance = 1e-6; self.max_iterations = 1; self.max_cpu_time = 0.0
        self.verbose = True; self.debug = False; self.print_solver_header = True; self.print_solver_options = False
        self.use_ilqr = True; self.enable_parallel = False; self.num_threads = 1; self.return_iteration_info = False
        self.warm_start = False; self.termination_scaling_max_factor = 100.0
        self.line_search = LineSearchOptions(); self.regularization = RegularizationOptions(); self.box_qp = BoxQPOptions()
        self.filter = FilterOptions(); self.log_barrier = LogBarrierOptions(); self.ipddp = IPDDPOptions()
        self.msipddp = MSIPDDPOptions()

    def to_pod(self, msipddp=False):
        """The POD of include/cddp_hip.h.  msipddp=True: the InteriorPointOptions half of options.msipddp (slack / dual init scale, barrier)
        travels in the ipddp_* / barrier_* fields, as the header documents for solver = MSIPDDP."""
        o = _api().default_options()
        o.tolerance = self.tolerance; o.acceptable_tolerance = self.acceptable_tolerance
        o.max_iterations = int(self.max_iterations); o.max_cpu_time = float(self.max_cpu_time); o.use_ilqr = 1 if self.use_ilqr else 0
        o.enable_parallel = 1 if self.enable_parallel else 0
        o.return_iteration_info = 1 if self.return_iteration_info else 0; o.warm_start = 1 if self.warm_start else 0
        o.termination_scaling_max_factor = self.termination_scaling_max_factor
        ls, rg, bq, fl, ip = self.line_search, self.regularization, self.box_qp, self.filter, (self.msipddp if msipddp else self.ipddp)
        o.ls_max_iterations = int(ls.max_iterations); o.ls_initial_step_size = ls.initial_step_size
        o.ls_min_step_size = ls.min_step_size; o.ls_step_reduction_factor = ls.step_reduction_factor
        o.reg_initial_value = rg.initial_value; o.reg_update_factor = rg.update_factor
        o.reg_max_value = rg.max_value; o.reg_min_value = rg.min_value
        o.boxqp_max_iterations = int(bq.max_iterations); o.boxqp_min_gradient_norm = bq.min_gradient_norm
        o.boxqp_min_relative_improvement = bq.min_relative_improvement; o.boxqp_step_decrease_factor = bq.step_decrease_factor
        o.boxqp_min_step_size = bq.min_step_size; o.boxqp_armijo_constant = bq.armijo_constant
        o.filter_merit_acceptance_threshold = fl.merit_acceptance_threshold
        o.filter_violation_acceptance_threshold = fl.violation_acceptance_threshold
        o.filter_max_violation_threshold = fl.max_violation_threshold
        o.filter_min_violation_for_armijo_check = fl.min_violation_for_armijo_check
        o.filter_armijo_constant = fl.armijo_constant
        o.ipddp_dual_var_init_scale = ip.dual_var_init_scale; o.ipddp_slack_var_init_scale = ip.slack_var_init_scale
        b = ip.barrier
        o.barrier_mu_initial = b.mu_initial; o.barrier_mu_min_value = b.mu_min_value
        o.barrier_mu_update_factor = b.mu_update_factor; o.barrier_mu_update_power = b.mu_update_power
        o.barrier_min_fraction_to_boundary = b.min_fraction_to_boundary; o.barrier_strategy = int(b.strategy)
        lb = self.log_barrier
        o.logddp_mu_initial = lb.barrier.mu_initial; o.logddp_mu_min_value = lb.barrier.mu_min_value
        o.logddp_mu_update_factor = lb.barrier.mu_update_factor; o.logddp_relaxed_delta = lb.relaxed_log_barrier_delta
        ms = self.msipddp
        o.msipddp_costate_var_init_scale = ms.costate_var_init_scale; o.msipddp_segment_length = int(ms.segment_length)
        o.msipddp_rollout_type = {"nonlinear": 0, "hybrid": 2}.get(ms.rollout_type, 1)
        o.msipddp_use_controlled_rollout = 1 if ms.use_controlled_rollout else 0
        return o


# ------------------------------------------------------------------------------------------------ plants
_INTEGRATORS = {"euler": 0, "heun": 1, "rk3": 2, "rk4": 3}


class DynamicalSystem:
    """dynamical_system.hpp / bind_dynamics.cpp:31-131.  The built-in plants below are descriptors of device kernels (`model` id);
    a Python subclass (model None) overrides get_continuous_dynamics (+ Jacobians) and runs through the host plug-in solve."""
    model = None

    def __init__(self, state_dim, control_dim, timestep, integration_type="euler"):
        if integration_type not in _INTEGRATORS:
            raise ValueError("Unknown integration type: " + str(integration_type))
        self.state_dim, self.control_dim, self.timestep, self.integration_type = state_dim, control_dim, timestep, integration_type
        self.params = []; self.lti_A = None; self.lti_B = None

    # -- the virtual interface (snake_case names of the pybind layer)
    def get_continuous_dynamics(self, state, control, time=0.0):
        raise RuntimeError("get_continuous_dynamics is not implemented for " + type(self).__name__)

    def get_discrete_dynamics(self, state, control, time=0.0):   # dynamical_system.cpp:28-83
        f, dt = self.get_continuous_dynamics, self.timestep
        x = np.asarray(state, dtype=np.float64); u = np.asarray(control, dtype=np.float64)
        it = self.integration_type
        k1 = np.asarray(f(x, u, time), dtype=np.float64)
        if it == "euler":
            return x + dt * k1
        if it == "heun":
            k2 = np.asarray(f(x + dt * k1, u, time + dt), dtype=np.float64)
            return x + (0.5 * dt) * (k1 + k2)
        if it == "rk3":
            k2 = np.asarray(f(x + (0.5 * dt) * k1, u, time + 0.5 * dt), dtype=np.float64)
            k3 = np.asarray(f((x - dt * k1) + (2 * dt) * k2, u, time + dt), dtype=np.float64)
            return x + (dt / 6) * ((k1 + 4.0 * k2) + k3)
        k2 = np.asarray(f(x + (0.5 * dt) * k1, u, time + 0.5 * dt), dtype=np.float64)
        k3 = np.asarray(f(x + (0.5 * dt) * k2, u, time + 0.5 * dt), dtype=np.float64)
        k4 = np.asarray(f(x + dt * k3, u, time + dt), dtype=np.float64)
        return x + (dt / 6) * (((k1 + 2.0 * k2) + 2.0 * k3) + k4)

    def _no_autodiff(self):   # bind_dynamics.cpp:35-45
        raise RuntimeError("Python-defined DynamicalSystem objects do not support getContinuousDynamicsAutodiff. Override "
                           "get_state_jacobian, get_control_jacobian, and any needed Hessian methods in Python, or use a built-in "
                           "C++ dynamics model.")

    def get_state_jacobian(self, state, control, time=0.0): self._no_autodiff()
    def get_control_jacobian(self, state, control, time=0.0): self._no_autodiff()
    def get_state_hessian(self, state, control, time=0.0): self._no_autodiff()
    def get_control_hessian(self, state, control, time=0.0): self._no_autodiff()
    def get_cross_hessian(self, state, control, time=0.0): self._no_autodiff()


class _BuiltinPlant(DynamicalSystem):
    """A plant the kernels implement (descriptor: model id + parameters).  Its host-side virtuals evaluate the SAME model source
    compiled for the host (cddp_hip_model_eval), so it can also be paired with Python objectives / constraints (plug-in solve)."""
    def _eval(self, state, control, want):
        api = _api()
        return api.model_eval(self.model, _INTEGRATORS[self.integration_type], self.timestep, self.params, self.state_dim, self.control_dim,
                              state, control, want=(want,))[want]
    def get_discrete_dynamics(self, state, control, time=0.0): return self._eval(state, control, "step")
    def get_continuous_dynamics(self, state, control, time=0.0):
        raise NotImplementedError("continuous dynamics of built-in plants are not exported; use get_discrete_dynamics / the Jacobians")
    def get_state_jacobian(self, state, control, time=0.0): return self._eval(state, control, "jac")[0]
    def get_control_jacobian(self, state, control, time=0.0): return self._eval(state, control, "jac")[1]
    def get_state_hessian(self, state, control, time=0.0): return list(self._eval(state, control, "hess")[0])
    def get_control_hessian(self, state, control, time=0.0): return list(self._eval(state, control, "hess")[1])
    def get_cross_hessian(self, state, control, time=0.0): return list(self._eval(state, control, "hess")[2])


class Pendulum(_BuiltinPlant):    # bind_dynamics.cpp:133-137
    def __init__(self, timestep, length=1.0, mass=1.0, damping=0.0, integration_type="euler"):
        self.model = _api().MODEL_PENDULUM
        super().__init__(2, 1, timestep, integration_type); self.params = [length, mass, damping, 9.81]


class CartPole(_BuiltinPlant):    # :152-158
    def __init__(self, timestep, integration_type="rk4", cart_mass=1.0, pole_mass=0.2, pole_length=0.5, gravity=9.81, damping=0.0):
        self.model = _api().MODEL_CARTPOLE
        super().__init__(4, 1, timestep, integration_type); self.params = [cart_mass, pole_mass, pole_length, gravity, damping]


class Unicycle(_BuiltinPlant):    # :139-141
    def __init__(self, timestep, integration_type="euler"):
        self.model = _api().MODEL_UNICYCLE
        super().__init__(3, 2, timestep, integration_type)


class Quadrotor(_BuiltinPlant):   # :177-182
    def __init__(self, timestep, mass, inertia_matrix, arm_length, integration_type="euler"):
        self.model = _api().MODEL_QUADROTOR
        super().__init__(13, 4, timestep, integration_type)
        J = np.asarray(inertia_matrix, dtype=np.float64)
        self.params = [mass, arm_length, J[0, 0], J[1, 1], J[2, 2], 9.81]


class HCW(_BuiltinPlant):           # spacecraft_linear.hpp:33: (timestep, mean_motion, mass, integration_type)
    def __init__(self, timestep, mean_motion, mass, integration_type="euler"):
        self.model = _api().MODEL_HCW
        super().__init__(6, 3, timestep, integration_type); self.params = [mean_motion, mass]


class Bicycle(_BuiltinPlant):       # bind_dynamics.cpp:143-146: (timestep, wheelbase, integration_type)
    def __init__(self, timestep, wheelbase, integration_type="euler"):
        self.model = _api().MODEL_BICYCLE
        super().__init__(4, 2, timestep, integration_type); self.params = [wheelbase]


class Car(_BuiltinPlant):           # bind_dynamics.cpp:148-150: (timestep, wheelbase, integration_type); a discrete plant (car.cpp:24-60)
    def __init__(self, timestep=0.03, wheelbase=2.0, integration_type="euler"):
        self.model = _api().MODEL_CAR
        super().__init__(4, 2, timestep, integration_type); self.params = [wheelbase]


class Manipulator(_BuiltinPlant):  # :189-191
    def __init__(self, timestep, integration_type="rk4"):
        self.model = _api().MODEL_MANIPULATOR
        super().__init__(6, 3, timestep, integration_type)


class LTISystem(DynamicalSystem):   # :233-237
    def __init__(self, A, B, timestep, integration_type="euler"):
        A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("A matrix must be square")
        if B.shape[0] != A.shape[0]:
            raise ValueError("B matrix must have same number of rows as A")
        self.model = _api().MODEL_LTI
        super().__init__(A.shape[0], B.shape[1], timestep, integration_type)
        self.lti_A, self.lti_B = A, B

    # host-side evaluation (lti_system.cpp:71-92), used when this plant is paired with a Python objective / constraint
    def get_discrete_dynamics(self, state, control, time=0.0): return self.lti_A @ np.asarray(state) + self.lti_B @ np.asarray(control)
    def get_state_jacobian(self, state, control, time=0.0): return (self.lti_A - np.eye(self.state_dim)) / self.timestep
    def get_control_jacobian(self, state, control, time=0.0): return self.lti_B / self.timestep
    def get_state_hessian(self, state, control, time=0.0): return [np.zeros((self.state_dim, self.state_dim))] * self.state_dim
    def get_control_hessian(self, state, control, time=0.0): return [np.zeros((self.control_dim, self.control_dim))] * self.state_dim
    def get_cross_hessian(self, state, control, time=0.0): return [np.zeros((self.control_dim, self.state_dim))] * self.state_dim


# ------------------------------------------------------------------------------------------------ objective / constraints
# ---- plants without device kernels: evaluated on the host and solved through the plug-in route (GPU backward passes, host rollouts)
def _fd_jacobian(f, x, h=2e-5):     # helper.hpp:95-119 (finite_difference_jacobian, central differences)
    x = np.asarray(x, dtype=np.float64); xp = x.copy(); cols = []
    for i in range(x.size):
        xp[i] = x[i] + h; fp = np.asarray(f(xp), dtype=np.float64)
        xp[i] = x[i] - h; fm = np.asarray(f(xp), dtype=np.float64)
        cols.append((fp - fm) / (2.0 * h)); xp[i] = x[i]
    return np.stack(cols, axis=1)


class _HostPlant(DynamicalSystem):
    """A plant of the reference restated on the host only (model None: no kernels).  Zero Hessian blocks unless the subclass has some."""
    def get_state_hessian(self, state, control, time=0.0): return [np.zeros((self.state_dim, self.state_dim)) for _ in range(self.state_dim)]
    def get_control_hessian(self, state, control, time=0.0): return [np.zeros((self.control_dim, self.control_dim)) for _ in range(self.state_dim)]
    def get_cross_hessian(self, state, control, time=0.0): return [np.zeros((self.control_dim, self.state_dim)) for _ in range(self.state_dim)]


class DubinsCar(_HostPlant):        # dubins_car.cpp:24-141 / bind_dynamics.cpp:160-162: state [x, y, theta], control [omega], constant speed
    def __init__(self, speed, timestep, integration_type="euler"):
        super().__init__(3, 1, timestep, integration_type); self.speed = float(speed)
    def get_continuous_dynamics(self, state, control, time=0.0):
        return np.array([self.speed * np.cos(state[2]), self.speed * np.sin(state[2]), control[0]])
    def get_state_jacobian(self, state, control, time=0.0):
        A = np.zeros((3, 3)); A[0, 2] = -self.speed * np.sin(state[2]); A[1, 2] = self.speed * np.cos(state[2]); return A
    def get_control_jacobian(self, state, control, time=0.0): B = np.zeros((3, 1)); B[2, 0] = 1.0; return B
    def get_state_hessian(self, state, control, time=0.0):
        H = super().get_state_hessian(state, control, time)
        H[0][2, 2] = -self.speed * np.cos(state[2]); H[1][2, 2] = -self.speed * np.sin(state[2]); return H


class DreyfusRocket(_HostPlant):    # dreyfus_rocket.cpp:24-83 / bind_dynamics.cpp:212-216: state [x, x_dot], control [theta]
    def __init__(self, timestep, integration_type="rk4", thrust_acceleration=64.0, gravity_acceleration=32.0):
        super().__init__(2, 1, timestep, integration_type)
        self.thrust_acceleration, self.gravity_acceleration = float(thrust_acceleration), float(gravity_acceleration)
    def get_thrust_acceleration(self): return self.thrust_acceleration
    def get_gravity_acceleration(self): return self.gravity_acceleration
    def get_continuous_dynamics(self, state, control, time=0.0):
        return np.array([state[1], self.thrust_acceleration * np.cos(control[0]) - self.gravity_acceleration])
    def get_state_jacobian(self, state, control, time=0.0): A = np.zeros((2, 2)); A[0, 1] = 1.0; return A
    def get_control_jacobian(self, state, control, time=0.0):
        B = np.zeros((2, 1)); B[1, 0] = -self.thrust_acceleration * np.sin(control[0]); return B
    def get_control_hessian(self, state, control, time=0.0):
        H = super().get_control_hessian(state, control, time); H[1][0, 0] = -self.thrust_acceleration * np.cos(control[0]); return H


def _cs_jacobian(f, x, h=1e-30):    # complex-step derivative: the value autodiff's forward duals give, to rounding (no subtraction, no step error)
    x = np.asarray(x, dtype=np.complex128); cols = []
    for i in range(x.size):
        xp = x.copy(); xp[i] += 1j * h
        cols.append(np.asarray(f(xp)).imag / h)
    return np.stack(cols, axis=1)


class Acrobot(_HostPlant):          # acrobot.cpp:24-96 / bind_dynamics.cpp:172-175: state [theta1, theta2, theta1_dot, theta2_dot], control [torque]
    """The reference differentiates its autodiff twin of the same expressions (dynamical_system.cpp: getStateJacobian by forward
    duals); here the Jacobians are complex-step derivatives of the one restatement below -- equal to the dual-number values to rounding."""
    gravity, friction = 9.81, 1.0     # acrobot.hpp:138-139
    def __init__(self, timestep, l1=1.0, l2=1.0, m1=1.0, m2=1.0, J1=1.0, J2=1.0, integration_type="euler"):
        super().__init__(4, 1, timestep, integration_type); self.l1, self.l2, self.m1, self.m2, self.J1, self.J2 = l1, l2, m1, m2, J1, J2
    def _f(self, s, c):
        l1, l2, m1, m2, J1, J2 = self.l1, self.l2, self.m1, self.m2, self.J1, self.J2
        th1, th2, w1, w2 = s[0], s[1], s[2], s[3]
        c1, s2, c2, c12 = np.cos(th1), np.sin(th2), np.cos(th2), np.cos(th1 + th2)
        m11 = m1 * l1 * l1 + J1 + m2 * (l1 * l1 + l2 * l2 + 2 * l1 * l2 * c2) + J2
        m12 = m2 * (l2 * l2 + l1 * l2 * c2) + J2
        m22 = l2 * l2 * m2 + J2
        tmp = l1 * l2 * m2 * s2
        b1 = -(2 * w1 * w2 + w2 * w2) * tmp; b2 = tmp * w1 * w1
        g1 = ((m1 + m2) * l1 * c1 + m2 * l2 * c12) * self.gravity; g2 = m2 * l2 * c12 * self.gravity
        r1 = 0.0 - b1 - g1 - self.friction * w1; r2 = c[0] - b2 - g2 - self.friction * w2
        inv_det = 1.0 / (m11 * m22 - m12 * m12)       # Eigen's 2 x 2 inverse: cofactors times 1 / det
        return np.array([w1, w2, (m22 * inv_det) * r1 + (-m12 * inv_det) * r2, (-m12 * inv_det) * r1 + (m11 * inv_det) * r2])
    def get_continuous_dynamics(self, state, control, time=0.0):
        return self._f(np.asarray(state, dtype=np.float64), np.asarray(control, dtype=np.float64))
    def get_state_jacobian(self, state, control, time=0.0):
        return _cs_jacobian(lambda s: self._f(s, np.asarray(control, dtype=np.complex128)), state)
    def get_control_jacobian(self, state, control, time=0.0):
        return _cs_jacobian(lambda c: self._f(np.asarray(state, dtype=np.complex128), c), control)
    def _no_hessian(self, *a, **k):
        raise NotImplementedError("Acrobot: second derivatives are not restated (the reference takes them from autodiff); use_ilqr = True")
    get_state_hessian = get_control_hessian = get_cross_hessian = _no_hessian


def _inv3_cofactor(M):              # Eigen's fixed 3 x 3 inverse: cofactors times 1 / det
    c = lambda i, j: M[(i + 1) % 3, (j + 1) % 3] * M[(i + 2) % 3, (j + 2) % 3] - M[(i + 1) % 3, (j + 2) % 3] * M[(i + 2) % 3, (j + 1) % 3]
    C = np.array([[c(i, j) for j in range(3)] for i in range(3)])
    det = M[0, 0] * C[0, 0] + M[0, 1] * C[0, 1] + M[0, 2] * C[0, 2]
    return C.T * (1.0 / det)


class Usv3Dof(_HostPlant):          # usv_3dof.cpp:11-110 / bind_dynamics.cpp: state [x, y, psi, u, v, r], control [tau_u, tau_v, tau_r]
    """Generic surface-vessel parameters of the reference (:17-34).  Jacobians: complex-step derivatives of the restated dynamics (the
    reference writes the same derivatives out by hand, :152-227); the control Hessian is zero (:237-248), the others are not restated."""
    def __init__(self, timestep, integration_type="euler"):
        super().__init__(6, 3, timestep, integration_type)
        self.m, self.Iz = 100.0, 10.0
        self.X_udot, self.Y_vdot, self.Y_rdot, self.N_vdot, self.N_rdot = -10.0, -50.0, -5.0, -5.0, -5.0
        self.X_u, self.Y_v, self.Y_r, self.N_v, self.N_r = -20.0, -100.0, 0.0, 0.0, -20.0
        M = np.diag([self.m, self.m, self.Iz]) + np.array([[-self.X_udot, 0, 0], [0, -self.Y_vdot, -self.Y_rdot], [0, -self.N_vdot, -self.N_rdot]])
        self.M_inv = _inv3_cofactor(M)
        self.D_L = np.array([[-self.X_u, 0, 0], [0, -self.Y_v, -self.Y_r], [0, -self.N_v, -self.N_r]])
    def _f(self, s, tau):
        psi, u, v, r = s[2], s[3], s[4], s[5]
        c, sn = np.cos(psi), np.sin(psi)
        m_x, m_y, m_yr = self.m - self.X_udot, self.m - self.Y_vdot, -self.Y_rdot
        nu = np.array([u, v, r])
        Cnu = np.array([(-m_y * v - m_yr * r) * r, (m_x * u) * r, (m_y * v + m_yr * r) * u + (-m_x * u) * v])
        nd = self.M_inv @ (tau - Cnu - self.D_L @ nu)
        return np.array([c * u - sn * v, sn * u + c * v, r, nd[0], nd[1], nd[2]])
    def get_continuous_dynamics(self, state, control, time=0.0):
        return self._f(np.asarray(state, dtype=np.float64), np.asarray(control, dtype=np.float64))
    def get_state_jacobian(self, state, control, time=0.0):
        return _cs_jacobian(lambda s: self._f(s, np.asarray(control, dtype=np.complex128)), state)
    def get_control_jacobian(self, state, control, time=0.0):
        return _cs_jacobian(lambda c: self._f(np.asarray(state, dtype=np.complex128), c), control)
    def _no_hessian(self, *a, **k):
        raise NotImplementedError("Usv3Dof: state / cross second derivatives are not restated (autodiff in the reference); use_ilqr = True")
    get_state_hessian = get_cross_hessian = _no_hessian


class SpacecraftLinearFuel(_HostPlant):   # spacecraft_linear_fuel.cpp:29-158 / bind_dynamics.cpp:199-203
    """HCW relative motion with mass: state [x, y, z, vx, vy, vz, mass, accumulated control effort], control [Fx, Fy, Fz]; the
    reference's Jacobians are central finite differences of the continuous dynamics (:124-141), its Hessians zero (:144-158)."""
    def __init__(self, timestep, mean_motion, isp, g0=9.80665, integration_type="euler"):
        super().__init__(8, 3, timestep, integration_type)
        self.mean_motion, self.isp, self.g0, self.epsilon = float(mean_motion), float(isp), float(g0), 1e-8
    def get_continuous_dynamics(self, state, control, time=0.0):
        x, y, z, vx, vy, vz, mass = (state[i] for i in range(7)); Fx, Fy, Fz = control[0], control[1], control[2]
        n = self.mean_motion; n2 = n * n
        t2 = Fx * Fx + Fy * Fy + Fz * Fz
        return np.array([vx, vy, vz, 2.0 * n * vy + 3.0 * n2 * x + Fx / mass, -2.0 * n * vx + Fy / mass, -n2 * z + Fz / mass,
                         -np.sqrt(t2 + self.epsilon) / (self.isp * self.g0), 0.5 * t2])
    def get_state_jacobian(self, state, control, time=0.0):
        return _fd_jacobian(lambda s: self.get_continuous_dynamics(s, control, time), state)
    def get_control_jacobian(self, state, control, time=0.0):
        return _fd_jacobian(lambda c: self.get_continuous_dynamics(state, c, time), control)


class Objective:                    # objective.hpp:30-120 / bind_objective.cpp:34-45: bound WITHOUT a constructor
    def __init__(self, *args, **kwargs):
        # pybind11's message for a class bound without py::init; Python-defined objectives derive from NonlinearObjective (:62-63)
        raise TypeError("pycddp.Objective: No constructor defined!")
    def running_cost(self, state, control, index): raise RuntimeError("running_cost is not implemented")
    def terminal_cost(self, final_state): raise RuntimeError("terminal_cost is not implemented")
    def evaluate(self, states, controls):
        total = 0.0
        for t in range(len(controls)):
            total += self.running_cost(states[t], controls[t], t)
        return total + self.terminal_cost(states[-1])


def _fd_gradient(f, x, h=2e-5):     # helper.hpp:34-55 (central differences)
    x = np.asarray(x, dtype=np.float64); g = np.zeros(x.size); xp = x.copy()
    for i in range(x.size):
        xp[i] = x[i] + h; fp = f(xp)
        xp[i] = x[i] - h; fm = f(xp)
        g[i] = (fp - fm) / (2.0 * h); xp[i] = x[i]
    return g


def _fd_hessian(f, x, h=2e-5):      # helper.hpp:158-179 (central differences of central-difference gradients)
    x = np.asarray(x, dtype=np.float64); H = np.zeros((x.size, x.size)); xp = x.copy()
    for i in range(x.size):
        xp[i] = x[i] + h; gp = _fd_gradient(f, xp, h)
        xp[i] = x[i] - h; gm = _fd_gradient(f, xp, h)
        H[:, i] = (gp - gm) / (2.0 * h); xp[i] = x[i]
    return H


class NonlinearObjective(Objective):   # objective.cpp:156-288: derivatives by finite differences unless overridden
    def __init__(self, timestep=0.1):
        self.timestep = timestep

    def running_cost(self, state, control, index): return 0.0
    def terminal_cost(self, final_state): return 0.0
    def get_running_cost_state_gradient(self, x, u, index): return _fd_gradient(lambda s: self.running_cost(s, u, index), x)
    def get_running_cost_control_gradient(self, x, u, index): return _fd_gradient(lambda c: self.running_cost(x, c, index), u)
    def get_final_cost_gradient(self, x): return _fd_gradient(lambda s: self.terminal_cost(s), x)
    def get_running_cost_state_hessian(self, x, u, index): return _fd_hessian(lambda s: self.running_cost(s, u, index), x)
    def get_running_cost_control_hessian(self, x, u, index): return _fd_hessian(lambda c: self.running_cost(x, c, index), u)
    def get_final_cost_hessian(self, x): return _fd_hessian(lambda s: self.terminal_cost(s), x, 2e-5)

    def get_running_cost_cross_hessian(self, x, u, index):   # objective.cpp:245-277: h = 2e-8, four-point stencil
        x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64); h = 2e-8
        M = np.zeros((u.size, x.size))
        for i in range(u.size):
            for j in range(x.size):
                sp = x.copy(); sp[j] += h; sm = x.copy(); sm[j] -= h; cp = u.copy(); cp[i] += h; cm = u.copy(); cm[i] -= h
                M[i, j] = (self.running_cost(sp, cp, index) - self.running_cost(sp, cm, index) - self.running_cost(sm, cp, index)
                           + self.running_cost(sm, cm, index)) / (4.0 * h * h)
        return M


class QuadraticObjective(Objective):   # objective.hpp: (Q, R, Qf, reference_state, reference_states, timestep)
    # host-side evaluation (objective.cpp:80-154), used when this objective meets a Python plant / constraint
    def _ref(self, index): return self.reference_states[index] if self.reference_states else self.reference_state
    def running_cost(self, x, u, index):
        e = np.asarray(x) - self._ref(index)
        return float(e @ (self.Q * self.timestep) @ e + np.asarray(u) @ (self.R * self.timestep) @ np.asarray(u))
    def terminal_cost(self, x):
        e = np.asarray(x) - self.reference_state
        return float(e @ self.Qf @ e)
    def get_running_cost_state_gradient(self, x, u, index): return 2.0 * (self.Q * self.timestep) @ (np.asarray(x) - self._ref(index))
    def get_running_cost_control_gradient(self, x, u, index): return 2.0 * (self.R * self.timestep) @ np.asarray(u)
    def get_final_cost_gradient(self, x): return 2.0 * self.Qf @ (np.asarray(x) - self.reference_state)
    def get_running_cost_state_hessian(self, x, u, index): return 2.0 * self.Q * self.timestep
    def get_running_cost_control_hessian(self, x, u, index): return 2.0 * self.R * self.timestep
    def get_running_cost_cross_hessian(self, x, u, index): return np.zeros((self.R.shape[0], self.Q.shape[0]))
    def get_final_cost_hessian(self, x): return 2.0 * self.Qf

    def __init__(self, Q, R, Qf, reference_state, reference_states=(), timestep=0.1):
        self.Q = np.asarray(Q, dtype=np.float64); self.R = np.asarray(R, dtype=np.float64); self.Qf = np.asarray(Qf, dtype=np.float64)
        for M_, n in ((self.Q, "Q"), (self.R, "R"), (self.Qf, "Qf")):
            if M_.ndim != 2 or M_.shape[0] != M_.shape[1]:
                raise ValueError(n + " matrix must be square")
        self.reference_state = np.asarray(reference_state, dtype=np.float64)
        self.reference_states = [np.asarray(r, dtype=np.float64) for r in reference_states]
        if self.reference_states and np.linalg.norm(self.reference_states[-1] - self.reference_state) > 1e-6:
            raise ValueError("Last reference state must be same as the reference state")   # objective.cpp:55-63
        self.timestep = timestep


def _pure_virtual(cls, name):       # what pybind11's PYBIND11_OVERRIDE_PURE raises when a Python subclass leaves the method out
    raise RuntimeError('Tried to call pure virtual function "%s::%s"' % (cls, name))


class Constraint:                   # constraint.hpp:31-142 / bind_constraints.cpp:9-117: the virtual interface a Python subclass overrides
    """Same surface as the pybind class: Constraint(name); evaluate / get_state_jacobian / get_control_jacobian / compute_violation
    (state, control, index=0), get_lower_bound, get_upper_bound, compute_violation_from_value(g), get_center, get_state_hessian /
    get_control_hessian / get_cross_hessian (lists of matrices, zero by default), get_dual_dim, name."""
    def __init__(self, name): self._name = str(name)
    name = property(lambda self: getattr(self, "_name", type(self).__name__))
    def get_dual_dim(self): return 0                                          # constraint.hpp:42
    def evaluate(self, state, control, index=0): _pure_virtual("Constraint", "evaluate")
    def get_lower_bound(self): _pure_virtual("Constraint", "get_lower_bound")
    def get_upper_bound(self): _pure_virtual("Constraint", "get_upper_bound")
    def get_state_jacobian(self, state, control, index=0): _pure_virtual("Constraint", "get_state_jacobian")
    def get_control_jacobian(self, state, control, index=0): _pure_virtual("Constraint", "get_control_jacobian")
    def compute_violation(self, state, control, index=0): _pure_virtual("Constraint", "compute_violation")
    def compute_violation_from_value(self, g): _pure_virtual("Constraint", "compute_violation_from_value")
    def get_center(self): raise RuntimeError("This constraint type does not have a center.")   # :86-89 (std::logic_error)
    def get_state_hessian(self, state, control, index=0):                      # :91-120: zero blocks by default
        n = np.asarray(state).size; return [np.zeros((n, n)) for _ in range(self.get_dual_dim())]
    def get_control_hessian(self, state, control, index=0):
        n = np.asarray(control).size; return [np.zeros((n, n)) for _ in range(self.get_dual_dim())]
    def get_cross_hessian(self, state, control, index=0):
        nx, nu = np.asarray(state).size, np.asarray(control).size; return [np.zeros((nu, nx)) for _ in range(self.get_dual_dim())]


class _BuiltinConstraint(Constraint):
    """Shared pieces of the built-in rows: -inf lower bounds, computeViolation = computeViolationFromValue(evaluate)."""
    def get_dual_dim(self): return int(np.asarray(self.get_upper_bound()).size)
    def get_lower_bound(self): return np.full(self.get_dual_dim(), -np.inf)
    def compute_violation(self, state, control, index=0): return self.compute_violation_from_value(self.evaluate(state, control, index))
    def compute_violation_from_value(self, g): return float(max(0.0, np.asarray(g, dtype=np.float64)[0]))


class ControlConstraint(_BuiltinConstraint):   # constraint.hpp:144-251 (BoxConstraint<Control>): g = [-u; u] s, upper = [-lb; ub] s
    _NAME = "ControlConstraint"
    def __init__(self, lower_bound, upper_bound, scale_factor=1.0):
        Constraint.__init__(self, self._NAME)
        self.lower = np.asarray(lower_bound, dtype=np.float64); self.upper = np.asarray(upper_bound, dtype=np.float64); self.scale = scale_factor
    def _v(self, x, u): return np.asarray(u, dtype=np.float64)
    def evaluate(self, x, u, index=0): v = self._v(x, u); return np.concatenate([-v, v]) * self.scale
    def get_upper_bound(self): return np.concatenate([-self.lower, self.upper]) * self.scale
    def get_state_jacobian(self, x, u, index=0): return np.zeros((2 * self.upper.size, np.asarray(x).size))
    def get_control_jacobian(self, x, u, index=0): n = self.upper.size; return np.vstack([-np.eye(n), np.eye(n)]) * self.scale
    def compute_violation_from_value(self, g):   # :237-240
        return float(np.maximum(np.asarray(g, dtype=np.float64) - self.get_upper_bound(), 0.0).sum())


class StateConstraint(ControlConstraint):
    _NAME = "StateConstraint"
    def _v(self, x, u): return np.asarray(x, dtype=np.float64)
    def get_state_jacobian(self, x, u, index=0): n = self.upper.size; return np.vstack([-np.eye(n), np.eye(n)]) * self.scale
    def get_control_jacobian(self, x, u, index=0): return np.zeros((2 * self.upper.size, np.asarray(u).size))


class BallConstraint(_BuiltinConstraint):   # constraint.hpp:313-404: g = -s |x[:d] - c|^2, upper = -s r^2
    def __init__(self, radius, center, scale_factor=1.0):
        Constraint.__init__(self, "BallConstraint")
        self.radius = radius; self.center = np.asarray(center, dtype=np.float64); self.scale = scale_factor
    def evaluate(self, x, u, index=0): dlt = np.asarray(x)[:self.center.size] - self.center; return np.array([-self.scale * float(dlt @ dlt)])
    def get_upper_bound(self): return np.array([-self.scale * self.radius * self.radius])
    def get_center(self): return self.center.copy()
    def compute_violation_from_value(self, g):   # :353-358: max(0, g - lower bound) with lower bound -inf, as the reference has it
        return float(max(0.0, float(np.asarray(g)[0]) - (-np.inf)))
    def get_state_jacobian(self, x, u, index=0):
        J = np.zeros((1, np.asarray(x).size)); J[0, :self.center.size] = -2.0 * self.scale * (np.asarray(x)[:self.center.size] - self.center); return J
    def get_control_jacobian(self, x, u, index=0): return np.zeros((1, np.asarray(u).size))
    def get_state_hessian(self, x, u, index=0):    # :387-396
        nx = np.asarray(x).size; d = self.center.size
        H = np.zeros((nx, nx)); H[:d, :d] = -2.0 * self.scale * np.eye(d)
        return [H]


class LinearConstraint(_BuiltinConstraint):   # constraint.hpp:253-311: g = A x, upper = b
    def __init__(self, A, b, scale_factor=1.0):
        Constraint.__init__(self, "LinearConstraint")
        self.A = np.asarray(A, dtype=np.float64); self.b = np.asarray(b, dtype=np.float64)
    def evaluate(self, x, u, index=0): return self.A @ np.asarray(x)
    def get_upper_bound(self): return self.b
    def get_state_jacobian(self, x, u, index=0): return self.A
    def get_control_jacobian(self, x, u, index=0): return np.zeros((self.b.size, np.asarray(u).size))
    def compute_violation_from_value(self, g): return float(max(0.0, float((self.b - np.asarray(g, dtype=np.float64)).max())))   # :302-305, as written


class SecondOrderConeConstraint(_BuiltinConstraint):   # constraint.hpp:626-800 / bind_constraints.cpp:146-150
    def __init__(self, cone_origin, opening_direction, cone_angle_fov, regularization_epsilon=1e-6, name="SecondOrderConeConstraint"):
        if cone_angle_fov < 0 or cone_angle_fov > np.pi:
            raise ValueError("SecondOrderConeConstraint: Cone angle must be between 0 and PI.")
        if regularization_epsilon <= 0:
            raise ValueError("SecondOrderConeConstraint: Regularization epsilon must be positive.")
        a = np.asarray(opening_direction, dtype=np.float64).reshape(3); n = float(np.sqrt(a @ a))
        if n == 0.0:
            raise ValueError("SecondOrderConeConstraint: Opening direction cannot be zero vector.")
        Constraint.__init__(self, name)
        self.origin = np.asarray(cone_origin, dtype=np.float64).reshape(3); self.axis = a / n
        self.fov = float(cone_angle_fov); self.cos_fov = float(np.cos(cone_angle_fov)); self.epsilon = float(regularization_epsilon)
    def get_dual_dim(self): return 1
    def evaluate(self, x, u, index=0):
        v = np.asarray(x, dtype=np.float64)[:3] - self.origin
        return np.array([np.sqrt(float(v @ v) + self.epsilon) * self.cos_fov - float(v @ self.axis)])
    def get_upper_bound(self): return np.zeros(1)
    def get_state_jacobian(self, x, u, index=0):
        x = np.asarray(x, dtype=np.float64); v = x[:3] - self.origin; rn = np.sqrt(float(v @ v) + self.epsilon)
        J = np.zeros((1, x.size)); J[0, :3] = self.cos_fov * (v / rn) - self.axis if rn > 1e-9 else -self.axis
        return J
    def get_control_jacobian(self, x, u, index=0): return np.zeros((1, np.asarray(u).size))
    def _no_hessian(self, *a, **k):              # constraint.hpp:783-800: std::logic_error -> the barrier drops the curvature term
        raise RuntimeError("SecondOrderConeConstraint: Hessians are not implemented.")
    get_state_hessian = get_control_hessian = get_cross_hessian = _no_hessian


def _norm_hessian(u, eps):          # constraint.hpp:899-920
    u = np.asarray(u, dtype=np.float64); term = float(u @ u) + eps; den = term ** 1.5
    return (term * np.eye(u.size) - np.outer(u, u)) / den if den > sys.float_info.min else np.zeros((u.size, u.size))


class ThrustMagnitudeConstraint(_BuiltinConstraint):   # constraint.hpp:802-927
    def __init__(self, min_thrust_norm, max_thrust_norm, epsilon=1e-6):
        if min_thrust_norm < 0.0:
            raise ValueError("ThrustMagnitudeConstraint: min_thrust_norm must be non-negative.")
        if max_thrust_norm < min_thrust_norm:
            raise ValueError("ThrustMagnitudeConstraint: max_thrust_norm must be greater than or equal to min_thrust_norm.")
        if epsilon <= 0.0:
            raise ValueError("ThrustMagnitudeConstraint: epsilon must be positive.")
        Constraint.__init__(self, "ThrustMagnitudeConstraint")
        self.min_norm, self.max_norm, self.epsilon = float(min_thrust_norm), float(max_thrust_norm), float(epsilon)
    def get_dual_dim(self): return 2
    def evaluate(self, x, u, index=0): n = np.sqrt(float(np.asarray(u) @ np.asarray(u))); return np.array([self.min_norm - n, n - self.max_norm])
    def get_upper_bound(self): return np.zeros(2)
    def compute_violation_from_value(self, g): return float(np.maximum(np.asarray(g, dtype=np.float64), 0.0).sum())
    def get_state_jacobian(self, x, u, index=0): return np.zeros((2, np.asarray(x).size))
    def get_control_jacobian(self, x, u, index=0):
        u = np.asarray(u, dtype=np.float64); rn = np.sqrt(float(u @ u) + self.epsilon); J = np.zeros((2, u.size))
        if not rn < self.epsilon:
            J[0] = -(u / rn); J[1] = u / rn
        return J
    def get_control_hessian(self, x, u, index=0): H = _norm_hessian(u, self.epsilon); return [-H, H]


class MaxThrustMagnitudeConstraint(_BuiltinConstraint):   # constraint.hpp:929-1048
    def __init__(self, max_thrust_norm, epsilon=1e-6):
        if max_thrust_norm < 0.0:
            raise ValueError("MaxThrustMagnitudeConstraint: max_thrust_norm must be non-negative.")
        if epsilon <= 0.0:
            raise ValueError("MaxThrustMagnitudeConstraint: epsilon must be positive.")
        Constraint.__init__(self, "MaxThrustMagnitudeConstraint")
        self.max_norm, self.epsilon = float(max_thrust_norm), float(epsilon)
    def get_dual_dim(self): return 1
    def evaluate(self, x, u, index=0): return np.array([np.sqrt(float(np.asarray(u) @ np.asarray(u))) - self.max_norm])
    def get_upper_bound(self): return np.zeros(1)
    def get_state_jacobian(self, x, u, index=0): return np.zeros((1, np.asarray(x).size))
    def get_control_jacobian(self, x, u, index=0):
        u = np.asarray(u, dtype=np.float64); rn = np.sqrt(float(u @ u) + self.epsilon); J = np.zeros((1, u.size))
        if rn > sys.float_info.min:
            J[0] = u / rn
        return J
    def get_control_hessian(self, x, u, index=0): return [_norm_hessian(u, self.epsilon)]


class TerminalEqualityConstraint:   # terminal_constraint.hpp:55-158: h(x_N) = x_N - target, Jacobian I
    def __init__(self, target_state):
        self.target = np.asarray(target_state, dtype=np.float64)

    def get_dual_dim(self): return int(self.target.size)
    def evaluate(self, final_state, control=None, index=0): return np.asarray(final_state, dtype=np.float64) - self.target
    def get_state_jacobian(self, final_state, control=None, index=0): return np.eye(self.target.size, np.asarray(final_state).size)


class TerminalInequalityConstraint:   # terminal_constraint.hpp:160-260: g_T(x_N) = A_N x_N - b_N <= 0
    def __init__(self, A, b):
        self.A = np.asarray(A, dtype=np.float64); self.b = np.asarray(b, dtype=np.float64)
        if self.A.shape[0] != self.b.size:
            raise ValueError("TerminalInequalityConstraint: A_N rows and b_N size mismatch.")

    def get_dual_dim(self): return int(self.A.shape[0])
    def evaluate(self, final_state, control=None, index=0): return self.A @ np.asarray(final_state, dtype=np.float64) - self.b
    def get_state_jacobian(self, final_state, control=None, index=0): return self.A


# ------------------------------------------------------------------------------------------------ solution
class SolutionHistory:              # bind_solver.cpp:523-540
    def __init__(self):
        self.objective = []; self.merit_function = []; self.step_length_primal = []; self.step_length_dual = []
        self.dual_infeasibility = []; self.primal_infeasibility = []; self.complementary_infeasibility = []
        self.barrier_mu = []; self.regularization = []


class CDDPSolution:                 # cddp_core.hpp:54-103 / bind_solver.cpp:542-570
    def __init__(self):
        self.solver_name = ""; self.status_message = ""; self.iterations_completed = 0; self.solve_time_ms = 0.0
        self.final_objective = 0.0; self.final_step_length = 1.0; self.final_regularization = 0.0
        self.time_points = []; self.state_trajectory = []; self.control_trajectory = []; self.feedback_gains = []
        self.final_primal_infeasibility = 0.0; self.final_dual_infeasibility = 0.0
        self.final_complementary_infeasibility = 0.0; self.final_barrier_mu = 0.0
        self.history = SolutionHistory()
        # NEW (no reference counterpart): which path solved the problem -- "resident" (device-resident batch kernels, shared straight-line
        # arithmetic) or "plugin" (host loop in the host libm + batched GPU backward passes); "" for an UnknownSolver result
        self.route = ""; self.arithmetic = ""


# ------------------------------------------------------------------------------------------------ CDDP
class CDDP:                         # cddp_core.hpp:214-423 / bind_solver.cpp:572-663
    def __init__(self, initial_state, reference_state, horizon, timestep, options=None):
        self._x0 = np.asarray(initial_state, dtype=np.float64).copy()
        self._xref = np.asarray(reference_state, dtype=np.float64).copy()
        self._N = int(horizon); self._dt = float(timestep)
        self._opt = options if options is not None else CDDPOptions()
        self._sys = None; self._obj = None; self._cons = {}; self._terms = {}
        self._X = None; self._U = None
        # NEW: route of LogDDP / MSIPDDP problems.  "auto": an eligible problem (built-in plant, nx <= 8, ...) runs on the resident kernels
        # (csrc/kernels_logddp.hpp, kernels_msipddp.hpp: the library's shared straight-line log / sin / cos) from BOTH solve() and
        # solve_batch() -- the same problem gets the same arithmetic, hence the same iteration count and status, whichever entry point is
        # used -- and everything else runs on the plug-in route (host loop in the host libm + stack-fed GPU sweeps);
        # "plugin" / "resident" force one.  The route taken is recorded in CDDPSolution.route / .arithmetic.
        self.msipddp_route = "auto"
        self.logddp_route = "auto"

    # -- setters (snake_case names of the pybind layer)
    def set_initial_state(self, x0): self._x0 = np.asarray(x0, dtype=np.float64).copy()
    def set_reference_state(self, xr): self._xref = np.asarray(xr, dtype=np.float64).copy()
    def set_reference_states(self, xs): self._xref_traj = [np.asarray(x, dtype=np.float64) for x in xs]
    def set_horizon(self, horizon): self._N = int(horizon)
    def set_timestep(self, timestep): self._dt = float(timestep)
    def set_options(self, options): self._opt = options
    def set_dynamical_system(self, system):
        if not isinstance(system, DynamicalSystem):
            raise TypeError("set_dynamical_system expects a DynamicalSystem (a built-in plant or a Python subclass)")
        if type(system) is DynamicalSystem:     # bind_solver.cpp:478-484
            raise TypeError("pycddp.DynamicalSystem is an abstract base class. Pass a concrete built-in model or a Python subclass that "
                            "implements the required methods.")
        self._sys = system
    def set_objective(self, objective):
        if not isinstance(objective, Objective):
            raise TypeError("set_objective expects an Objective (QuadraticObjective or a Python subclass)")
        self._obj = objective
    def add_constraint(self, name, constraint):
        if constraint is None:
            raise RuntimeError("Cannot add null constraint.")
        if type(constraint) is Constraint:      # bind_solver.cpp:504-510
            raise TypeError("pycddp.Constraint is an abstract base class. Pass a concrete built-in constraint or a Python subclass that "
                            "implements the required methods.")
        self._cons[name] = constraint
    def add_terminal_constraint(self, name, constraint):
        if constraint is None:
            raise RuntimeError("Cannot add null constraint.")
        self._terms[name] = constraint
    def remove_constraint(self, name): return self._cons.pop(name, None) is not None
    def remove_terminal_constraint(self, name): return self._terms.pop(name, None) is not None
    def set_initial_trajectory(self, X, U):
        X = [np.asarray(x, dtype=np.float64).reshape(-1) for x in X]; U = [np.asarray(u, dtype=np.float64).reshape(-1) for u in U]
        if self._sys is None:   # validateInitialTrajectory, bind_solver.cpp:106-152 (getStateDim throws without a system, cddp_core.cpp)
            raise ValueError("set_initial_trajectory failed while querying dimensions (is a dynamical system set?): "
                             "Dynamical system is not set")
        nx, nu = self._sys.state_dim, self._sys.control_dim
        if len(X) != self._N + 1 or len(U) != self._N:
            raise ValueError("set_initial_trajectory expected X length %d and U length %d, got X length %d and U length %d."
                             % (self._N + 1, self._N, len(X), len(U)))
        for i, x in enumerate(X):
            if x.size != nx:
                raise ValueError("set_initial_trajectory expected state vector %d to have dimension %d, got %d." % (i, nx, x.size))
        for i, u in enumerate(U):
            if u.size != nu:
                raise ValueError("set_initial_trajectory expected control vector %d to have dimension %d, got %d." % (i, nu, u.size))
        self._X, self._U = np.stack(X), np.stack(U)

    initial_state = property(lambda self: self._x0)
    reference_state = property(lambda self: self._xref)
    horizon = property(lambda self: self._N)
    timestep = property(lambda self: self._dt)
    state_dim = property(lambda self: self._sys.state_dim)
    control_dim = property(lambda self: self._sys.control_dim)
    options = property(lambda self: self._opt)

    # -- problem descriptor for the C-ABI
    def _problem(self, solver_kind):
        api = _api()
        if self._sys is None:
            raise RuntimeError("Dynamical system must be set before solving.")   # cddp_core.cpp:277-282
        if self._obj is None:
            raise RuntimeError("Objective function must be set before solving.")
        s, ob = self._sys, self._obj
        traj = ob.reference_states if ob.reference_states else None
        p = api.Problem(solver_kind, s.model, _INTEGRATORS[s.integration_type], s.state_dim, s.control_dim, self._N, self._dt,
                        ob.Q, ob.R, ob.Qf, ob.reference_state, model_params=s.params, lti_A=s.lti_A, lti_B=s.lti_B,
                        x_ref_traj=None if traj is None else np.stack(traj), options=self._opt.to_pod(msipddp=(solver_kind == api.SOLVER_MSIPDDP)))
        for name in sorted(self._cons):          # std::map order
            c = self._cons[name]
            if isinstance(c, StateConstraint): p.add_state_box(name, c.lower, c.upper, c.scale)
            elif isinstance(c, ControlConstraint): p.add_control_box(name, c.lower, c.upper, c.scale)
            elif isinstance(c, BallConstraint): p.add_ball(name, c.radius, c.center, c.scale)
            elif isinstance(c, LinearConstraint): p.add_linear(name, c.A, c.b)
            elif isinstance(c, SecondOrderConeConstraint): p.add_second_order_cone(name, c.origin, c.axis, c.fov, c.epsilon)
            elif isinstance(c, ThrustMagnitudeConstraint): p.add_thrust_magnitude(name, c.min_norm, c.max_norm, c.epsilon)
            elif isinstance(c, MaxThrustMagnitudeConstraint): p.add_max_thrust_magnitude(name, c.max_norm, c.epsilon)
            else: raise NotImplementedError("constraint type %s has no device kernel" % type(c).__name__)
        for name in sorted(self._terms):
            c = self._terms[name]
            if isinstance(c, TerminalEqualityConstraint): p.add_terminal_equality(name, c.target)
            elif isinstance(c, TerminalInequalityConstraint): p.add_terminal_inequality(name, c.A, c.b)
            else: raise NotImplementedError("terminal constraint type %s has no device kernel" % type(c).__name__)
        return p

    def _solve(self, name, x0s, resident_batch=False):
        api = _api()
        # LogDDP: a built-in plant with nx <= 8 runs on the resident LogDDP kernels (csrc/kernels_logddp.hpp: one device-resident batch in
        # the library's shared straight-line log / sin / cos) from solve() and solve_batch() alike; other problems, or logddp_route =
        # "plugin", take the host loop + stack-fed GPU sweeps (cddp_hip_plugin_solve: the host libm, the reference's own arithmetic)
        eligible = name == "LogDDP" and self._sys is not None and not self._needs_host_plugins() and self._sys.state_dim <= 8
        if name == "LogDDP" and self.logddp_route == "resident" and not eligible:
            raise NotImplementedError("the resident LogDDP kernels serve built-in plants with nx <= 8 and built-in objective / constraints")
        resident_logddp = eligible and self.logddp_route != "plugin"
        if name == "LogDDP" and not resident_logddp:
            return self._solve_plugins(name, api.SOLVER_LOGDDP, x0s)
        # MSIPDDP: same split (round 4, csrc/kernels_msipddp.hpp): solve_batch() of a built-in plant with nx <= 8, no terminal set and --
        # with path constraints -- nu = 1 or nx = nu (the shapes msipddp_solver.cpp:1398 defines) is one device-resident batch
        # (round 5: the library accepts nx <= 13 -- the unconstrained quadrotor -- on one-lane, scratch-backed sweeps: "auto" keeps such a problem on
        #  the plug-in route, msipddp_route = "resident" asks for the device)
        ms_nx_cap = 13 if self.msipddp_route == "resident" else 8
        ms_eligible = (name == "MSIPDDP" and self._sys is not None and not self._needs_host_plugins() and self._sys.state_dim <= ms_nx_cap and not self._terms
                       and (not self._cons or self._sys.control_dim == 1 or self._sys.state_dim == self._sys.control_dim))
        if name == "MSIPDDP" and self.msipddp_route == "resident" and not ms_eligible:
            raise NotImplementedError("the resident MSIPDDP kernels serve built-in plants with nx <= 13, built-in objective / constraints, no terminal set, and nu = 1 or nx = nu once a path constraint is present")
        resident_msipddp = ms_eligible and self.msipddp_route != "plugin"
        if name == "MSIPDDP" and not resident_msipddp:   # path constraints only for nu = 1 or nx = nu (msipddp_solver.cpp:1398, the library says so)
            return self._solve_plugins(name, api.SOLVER_MSIPDDP, x0s)
        if name not in ("CLDDP", "IPDDP", "LogDDP", "MSIPDDP"):
            sol = CDDPSolution()                 # cddp_core.cpp:243-265: unknown names do not throw
            sol.solver_name = name; sol.status_message = "UnknownSolver - No solver registered for '%s'" % name
            return [sol for _ in range(len(x0s))]
        kind = api.SOLVER_IPDDP if name == "IPDDP" else api.SOLVER_LOGDDP if name == "LogDDP" else api.SOLVER_MSIPDDP if name == "MSIPDDP" else api.SOLVER_CLDDP
        if self._needs_host_plugins():
            return self._solve_plugins(name, kind, x0s)
        p = self._problem(kind)
        B = len(x0s)
        x0 = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.float64) for x in x0s]))
        U0 = None if self._U is None else np.ascontiguousarray(np.tile(self._U, (B, 1, 1)))
        X0 = None if self._X is None else np.ascontiguousarray(np.tile(self._X, (B, 1, 1)))
        hs = api.HipBatchSolver(p, B)
        try:
            hs.set_initial(x0, U0, X0)
            st = hs.solve()
            res = hs.results(); X, U = hs.trajectory(); K, _ = hs.gains()
            hist = hs.history(min(B, 64)) if self._opt.return_iteration_info else None
        finally:
            hs.close()
        out = []
        for b in range(B):
            s = CDDPSolution()
            s.solver_name = name; s.status_message = api.STATUS_STRINGS[int(res["status"][b])]
            s.route = "resident"; s.arithmetic = "device, shared straight-line sin / cos / log (csrc/dev_trig.hpp)"
            s.iterations_completed = int(res["iterations"][b]); s.solve_time_ms = float(st.solve_ms)
            s.final_objective = float(res["final_objective"][b]); s.final_step_length = float(res["alpha_pr"][b])
            s.final_regularization = float(res["regularization"][b])
            s.final_primal_infeasibility = float(res["inf_pr"][b]); s.final_dual_infeasibility = float(res["inf_du"][b])
            s.final_complementary_infeasibility = float(res["inf_comp"][b]); s.final_barrier_mu = float(res["barrier_mu"][b])
            s.time_points = [t * self._dt for t in range(self._N + 1)]
            s.state_trajectory = [X[b, t].copy() for t in range(self._N + 1)]
            s.control_trajectory = [U[b, t].copy() for t in range(self._N)]
            s.feedback_gains = [K[b, t].copy() for t in range(self._N)]
            if hist is not None and b < len(hist):
                h = hist[b]
                s.history.objective = list(h[:, 0]); s.history.merit_function = list(h[:, 1])
                s.history.step_length_primal = list(h[:, 2]); s.history.step_length_dual = list(h[:, 3])
                s.history.dual_infeasibility = list(h[:, 4]); s.history.primal_infeasibility = list(h[:, 5])
                s.history.complementary_infeasibility = list(h[:, 6])
                s.history.barrier_mu = list(h[:, 7]) if name in ("IPDDP", "LogDDP", "MSIPDDP") else []   # logddp_solver.cpp:278-284
                s.history.regularization = list(h[:, 8])
            out.append(s)
        return out

    # -- host plug-in path: Python subclasses of DynamicalSystem / Objective / Constraint (cddp_hip_plugin_solve)
    def _needs_host_plugins(self):
        if self._sys is None or self._obj is None:
            return False
        builtin_con = (ControlConstraint, StateConstraint, BallConstraint, LinearConstraint, SecondOrderConeConstraint,
                       ThrustMagnitudeConstraint, MaxThrustMagnitudeConstraint)
        return (self._sys.model is None or type(self._obj) is not QuadraticObjective
                or any(type(c) not in builtin_con for c in self._cons.values()))

    def _solve_plugins(self, name, kind, x0s):
        api = _api()
        s, ob = self._sys, self._obj
        nx, nu, N, dt = s.state_dim, s.control_dim, self._N, self._dt
        names = sorted(self._cons)          # std::map order
        ipddp = kind in (api.SOLVER_IPDDP, api.SOLVER_LOGDDP, api.SOLVER_MSIPDDP)
        cons = [self._cons[n] for n in names] if ipddp else []
        dims = [int(c.get_dual_dim()) for c in cons]
        lo = up = None
        if not ipddp and "ControlConstraint" in self._cons and isinstance(self._cons["ControlConstraint"], ControlConstraint):
            lo, up = self._cons["ControlConstraint"].lower, self._cons["ControlConstraint"].upper   # clddp_solver.cpp:85-86

        def constraints(x, u, index, want):   # evaluate / Jacobians with the step index, as the solvers call them (constraint.hpp:44-72)
            g = np.concatenate([np.asarray(c.evaluate(x, u, index), dtype=np.float64) - np.asarray(c.get_upper_bound(), dtype=np.float64) for c in cons])
            if not want:
                return g, None, None
            return (g, np.vstack([np.asarray(c.get_state_jacobian(x, u, index), dtype=np.float64).reshape(-1, nx) for c in cons]),
                    np.vstack([np.asarray(c.get_control_jacobian(x, u, index), dtype=np.float64).reshape(-1, nu) for c in cons]))

        def constraint_hessians(x, u, index):   # getHessians (constraint.hpp:122-131); a constraint that throws contributes no curvature
            out = ([], [], [])
            for c, r in zip(cons, dims):
                try:
                    trip = (c.get_state_hessian(x, u, index), c.get_control_hessian(x, u, index), c.get_cross_hessian(x, u, index))
                    trip = tuple(np.asarray(np.stack([np.asarray(h, dtype=np.float64) for h in hl]) if len(hl) else np.zeros((0, 1, 1))) for hl in trip)
                except RuntimeError:
                    trip = (np.zeros((r, nx, nx)), np.zeros((r, nu, nu)), np.zeros((r, nu, nx)))
                for k, shape in enumerate(((r, nx, nx), (r, nu, nu), (r, nu, nx))):
                    out[k].append(trip[k].reshape(shape))
            return tuple(np.concatenate(o) for o in out)

        def hessians(x, u, t):
            return (np.stack([np.asarray(h) for h in s.get_state_hessian(x, u, t)]), np.stack([np.asarray(h) for h in s.get_control_hessian(x, u, t)]),
                    np.stack([np.asarray(h) for h in s.get_cross_hessian(x, u, t)]))

        # terminal set (round 6): only IPDDP reads it (ipddp_solver.cpp:84-215); objects in std::map order, equality = TerminalEqualityConstraint
        tnames = sorted(self._terms) if kind == api.SOLVER_IPDDP else []
        tcons = [self._terms[n] for n in tnames]
        tdims = [int(c.get_dual_dim()) for c in tcons]
        teq = [1 if isinstance(c, TerminalEqualityConstraint) else 0 for c in tcons]

        def terminal(xN, want):   # evaluateTerminal{Equality,Inequality}Residual / Jacobian (:118-201): evaluate(x_N) and getStateJacobian(x_N)
            r = np.concatenate([np.asarray(c.evaluate(xN), dtype=np.float64).reshape(-1) for c in tcons])
            if not want:
                return r, None
            return r, np.vstack([np.asarray(c.get_state_jacobian(xN), dtype=np.float64).reshape(-1, nx) for c in tcons])

        B = len(x0s)
        x0 = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.float64) for x in x0s]))
        U0 = None if self._U is None else np.ascontiguousarray(np.tile(self._U, (B, 1, 1)))
        X0 = None if self._X is None else np.ascontiguousarray(np.tile(self._X, (B, 1, 1)))
        import time as _time
        t0 = _time.perf_counter()
        res, X, U, K = api.plugin_solve(
            kind, nx, nu, N, dt, self._opt.to_pod(msipddp=(kind == api.SOLVER_MSIPDDP)), x0, U0, X0,
            discrete_dynamics=lambda x, u, t: s.get_discrete_dynamics(x, u, t),
            jacobians=(lambda x, u, t: s._eval(x, u, "jac")) if isinstance(s, _BuiltinPlant) else
                      (lambda x, u, t: (s.get_state_jacobian(x, u, t), s.get_control_jacobian(x, u, t))),
            hessians=None if self._opt.use_ilqr else hessians,
            running_cost=lambda x, u, i: ob.running_cost(x, u, i), terminal_cost=lambda x: ob.terminal_cost(x),
            running_cost_derivatives=lambda x, u, i: (ob.get_running_cost_state_gradient(x, u, i), ob.get_running_cost_control_gradient(x, u, i),
                                                      ob.get_running_cost_state_hessian(x, u, i), ob.get_running_cost_control_hessian(x, u, i),
                                                      ob.get_running_cost_cross_hessian(x, u, i)),
            terminal_cost_derivatives=lambda x: (ob.get_final_cost_gradient(x), ob.get_final_cost_hessian(x)),
            constraints=constraints if cons else None, constraint_dims=dims, control_lower=lo, control_upper=up,
            constraint_hessians=constraint_hessians if (cons and (kind == api.SOLVER_LOGDDP or (kind == api.SOLVER_MSIPDDP and not self._opt.use_ilqr))) else None,
            terminal=terminal if tcons else None, terminal_dims=tdims, terminal_equality=teq)
        ms = (_time.perf_counter() - t0) * 1e3
        out = []
        for b in range(B):
            sol = CDDPSolution()
            sol.solver_name = name; sol.status_message = api.STATUS_STRINGS[int(res["status"][b])]
            sol.route = "plugin"; sol.arithmetic = "host libm (plug-in callbacks and outer loop on the host, batched GPU backward passes)"
            sol.iterations_completed = int(res["iterations"][b]); sol.solve_time_ms = ms
            sol.final_objective = float(res["final_objective"][b]); sol.final_step_length = float(res["alpha_pr"][b])
            sol.final_regularization = float(res["regularization"][b])
            sol.final_primal_infeasibility = float(res["inf_pr"][b]); sol.final_dual_infeasibility = float(res["inf_du"][b])
            sol.final_complementary_infeasibility = float(res["inf_comp"][b]); sol.final_barrier_mu = float(res["barrier_mu"][b])
            sol.time_points = [t * dt for t in range(N + 1)]
            sol.state_trajectory = [X[b, t].copy() for t in range(N + 1)]
            sol.control_trajectory = [U[b, t].copy() for t in range(N)]
            sol.feedback_gains = [K[b, t].copy() for t in range(N)]
            out.append(sol)
        return out

    def solve(self, solver_type=SolverType.CLDDP):
        name = solver_type.value if isinstance(solver_type, SolverType) else str(solver_type)
        sol = self._solve(name, [self._x0])[0]
        if sol.state_trajectory:     # the context keeps the solution, as the reference solvers leave it (cddp_solver_base.cpp:161-171)
            self._X = np.stack(sol.state_trajectory); self._U = np.stack(sol.control_trajectory)
        return sol

    def solve_by_name(self, solver_name):   # bind_solver.cpp:95-100, 637-654; aliases of cddp_core.cpp:221-230
        alias = {"CLCDDP": "CLDDP", "LOGDDP": "LogDDP"}
        if solver_name not in ("CLDDP", "CLCDDP", "LogDDP", "LOGDDP", "IPDDP", "MSIPDDP"):
            raise ValueError("Unknown solver '%s'." % solver_name)
        return self.solve(alias.get(solver_name, solver_name))

    def solve_batch(self, x0s, solver_type=SolverType.IPDDP):
        """One device-resident batch: solution i starts from x0s[i] (same problem, same initial trajectory guess)."""
        name = solver_type.value if isinstance(solver_type, SolverType) else str(solver_type)
        return self._solve(name, list(x0s), resident_batch=True)


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("enum", "importlib", "os", "sys", "np")]
__version__ = "0.1.0"
