"""The reference-side adapter (integration/hip_batch_solver.cpp) cannot be compiled in this image (it includes the REAL cddp-cpp headers,
which need Eigen 3.4 / autodiff 1.1.2).  What CAN be checked here: every C-ABI entry point, struct field and enumerator it uses exists in
include/cddp_hip.h with that spelling; the recipe that compiles it refuses loudly without the dependencies; every reference header /
accessor it relies on exists in the reference checkout when one is present (skipped on the GPU box, where /root/reference does not exist)."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(REPO, "integration", "hip_batch_solver.cpp")).read()
HDR = open(os.path.join(REPO, "include", "cddp_hip.h")).read()


def test_every_c_abi_name_the_adapter_uses_is_declared():
    used = set(re.findall(r"\b(cddp_hip_[a-z_0-9]+)\b", SRC)) | set(re.findall(r"\b(CDDP_HIP_[A-Z_0-9]+)\b", SRC))
    used -= {"CDDP_HIP_REFERENCE_HAS_GETTERS", "CDDP_HIP_F4_ROUTE"}     # the adapter's own build switch / an environment variable name
    missing = sorted(n for n in used if not re.search(r"\b%s\b" % re.escape(n), HDR))
    assert not missing, missing


def test_every_struct_field_the_adapter_writes_exists():
    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), HDR, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return set(re.findall(r"\b([a-z_][a-z_0-9]*)\s*(?:\[[^\]]*\])?\s*[;,)]", body)) | set(re.findall(r"\(\*([a-z_]+)\)", body))
    for var, struct in (("o", "cddp_hip_options"), ("p", "cddp_hip_problem"), ("pl", "cddp_hip_plugin")):
        have = fields(struct)
        wrote = set(re.findall(r"\b%s\.([a-z_][a-z_0-9]*)\s*=" % var, SRC))
        assert wrote, struct
        assert wrote <= have, (struct, sorted(wrote - have))
    have = fields("cddp_hip_constraint") | fields("cddp_hip_terminal_constraint")
    wrote = set(re.findall(r"\bd\.([a-z_][a-z_0-9]*)\s*=", SRC)) - {"x_ref", "x_ref_traj"}   # (ObjectiveDesc's own members share the variable name)
    assert wrote <= have, sorted(wrote - have)
    have = fields("cddp_hip_result")
    read = set(re.findall(r"\br\.([a-z_][a-z_0-9]*)\b", SRC)) - {"data", "size"}   # (`r` is also an Eigen vector / std::vector in two callbacks)
    assert read <= have, sorted(read - have)


def test_build_recipe_refuses_without_the_pinned_dependencies():
    env = {k: v for k, v in os.environ.items() if k not in ("EIGEN3_INCLUDE_DIR", "AUTODIFF_INCLUDE_DIR")}
    r = subprocess.run(["bash", os.path.join(REPO, "oracle", "ref_pin", "build_ref.sh")], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and ("EIGEN3_INCLUDE_DIR" in r.stderr or "reference sources not found" in r.stderr), (r.returncode, r.stderr)
    recipe = open(os.path.join(REPO, "oracle", "ref_pin", "build_ref.sh")).read()
    assert "integration/hip_batch_solver.cpp" in recipe and "integration/test_hip_registry.cpp" in recipe


REF = "/root/reference/include/cddp-cpp"


@pytest.mark.skipif(not os.path.isdir(REF), reason="no reference checkout on this machine")
def test_reference_headers_and_accessors_the_adapter_relies_on_exist():
    for inc in re.findall(r'#include "((?:cddp_core|dynamics_model)/[a-z_]+\.hpp)"', SRC):
        assert os.path.isfile(os.path.join(REF, inc)), inc
    text = {f: open(os.path.join(REF, f)).read() for f in ("cddp_core/cddp_core.hpp", "cddp_core/constraint.hpp", "cddp_core/terminal_constraint.hpp",
                                                           "cddp_core/objective.hpp", "cddp_core/dynamical_system.hpp", "cddp_core/options.hpp",
                                                           "dynamics_model/pendulum.hpp", "dynamics_model/cartpole.hpp", "dynamics_model/lti_system.hpp")}
    need = {"cddp_core/cddp_core.hpp": ["registerSolver", "getConstraintSet", "getTerminalConstraintSet", "getInitialState", "getStateDim", "getControlDim",
                                        "getHorizon", "getTimestep", "getOptions", "getSystem", "getObjective", "X_", "U_", "cost_", "alpha_pr_", "regularization_",
                                        "inf_pr_", "inf_du_", "inf_comp_", "feedback_gains", "final_barrier_mu", "step_length_primal"],
            "cddp_core/constraint.hpp": ["rawLowerBound", "rawUpperBound", "getCenter", "getRadius", "getDualDim", "getUpperBound", "getStateJacobian",
                                         "getControlJacobian", "getStateHessian", "using ControlConstraint", "using StateConstraint", "class LinearConstraint", "class BallConstraint"],
            "cddp_core/terminal_constraint.hpp": ["class TerminalEqualityConstraint", "class TerminalInequalityConstraint"],
            "cddp_core/objective.hpp": ["getQ()", "getR()", "getQf()", "getReferenceState", "getReferenceStates", "getRunningCostStateGradient", "getFinalCostHessian"],
            "cddp_core/dynamical_system.hpp": ["getIntegrationType", "getDiscreteDynamics", "getStateJacobian", "getControlJacobian", "getCrossHessian"],
            "cddp_core/options.hpp": ["barrier_update_dual_weight", "warmstart_interior_factor", "relaxed_log_barrier_delta", "costate_var_init_scale", "min_fraction_to_boundary"],
            "dynamics_model/pendulum.hpp": ["getLength", "getMass", "getDamping", "getGravity"],
            "dynamics_model/cartpole.hpp": ["getCartMass", "getPoleMass", "getPoleLength", "getGravity", "getDamping"],
            "dynamics_model/lti_system.hpp": ["getA()", "getB()"]}
    for f, names in need.items():
        for n in names:
            assert n in text[f], (f, n)
    # every options field the adapter reads exists under that name in options.hpp
    for fld in set(re.findall(r"\bc\.(?:line_search|regularization|box_qp|filter|ipddp|msipddp|log_barrier)(?:\.barrier)?\.([a-z_0-9]+)", SRC)):
        assert re.search(r"\b%s\b" % fld, text["cddp_core/options.hpp"] + open(os.path.join(REF, "cddp_core/boxqp.hpp")).read()), fld
