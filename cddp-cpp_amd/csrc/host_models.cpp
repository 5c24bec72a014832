// Host evaluation of the built-in plants: dev_models.hpp (the source the kernels compile for gfx950) compiled for the host.
//
// Why: the plug-in solve (plugin_solve.hip) runs its forward passes on the host through the DynamicalSystem callbacks.  A problem
// that pairs a BUILT-IN plant with a user Objective / Constraint subclass -- the shape of the reference's own car-parking and
// NonlinearObjective tests (tests/cddp_core/test_ipddp_solver.cpp:628-885, python/tests/test_nonlinear_objective.py) -- needs
// DynamicalSystem::getDiscreteDynamics / getStateJacobian / ... of that plant on the host.  One source, two targets: no second
// restatement to drift.  (The CPU checker under the repo's test tree has its own, independently written plants; the tests compare.)
//
// Built with g++ -ffp-contract=off (Makefile); sin / cos are the host libm's by default, the shared straight-line routines under
// CDDP_TRIG_SHARED -- the same switch as the device objects of the same library.
#define CDDP_HOST_MODELS 1
#define CDDP_TRIG_HOST 1
#define DEV inline
#include "dev_models.hpp"

#include <string>

namespace {
using namespace cddp_dev;

template <class Model, bool H = Model::kHasHess> struct HessOf {
  static bool run(const double *p, const double *x, const double *u, double *fxx, double *fuu, double *fux) { Model::hess(p, x, u, fxx, fuu, fux); return true; }
};
template <class Model> struct HessOf<Model, false> {
  static bool run(const double *, const double *, const double *, double *, double *, double *) { return false; }
};

template <class Model>
int eval(int integrator, double dt, const double *params, int nx, int nu, const double *x, const double *u, double *x_next, double *fx, double *fu,
         double *fxx, double *fuu, double *fux, std::string &err) {
  if (nx != Model::NX || nu != Model::NU) { err = "model dimensions do not match the plant"; return -2; }
  double p[32];
  for (int i = 0; i < 32; ++i) p[i] = i < CDDP_HIP_MAX_MODEL_PARAMS ? params[i] : 0.0;
  if (x_next) Stepper<Model>::step(integrator, dt, p, x, u, x_next);
  if (fx || fu) {
    double Fx[Model::NX * Model::NX], Fu[Model::NX * Model::NU];
    Model::jac(p, x, u, Fx, Fu);
    if (fx) for (int i = 0; i < Model::NX * Model::NX; ++i) fx[i] = Fx[i];
    if (fu) for (int i = 0; i < Model::NX * Model::NU; ++i) fu[i] = Fu[i];
  }
  if (fxx || fuu || fux) {
    if (!fxx || !fuu || !fux) { err = "Hessian outputs come as a triple"; return -2; }
    if (!HessOf<Model>::run(p, x, u, fxx, fuu, fux)) { err = "this plant has no Hessian tensors (options.use_ilqr = 0 is not available for it)"; return -3; }
  }
  return 0;
}
}  // namespace

// C++ linkage: called from capi.hip (cddp_hip_model_eval), which owns the error string
int cddp_host_model_eval(int model, int integrator, double dt, const double *params, int nx, int nu, const double *x, const double *u, double *x_next,
                         double *fx, double *fu, double *fxx, double *fuu, double *fux, std::string &err) {
  if (integrator < CDDP_HIP_EULER || integrator > CDDP_HIP_RK4) { err = "Integration type not supported!"; return -2; }
  switch (model) {
    case CDDP_HIP_MODEL_PENDULUM: return eval<PendulumModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_CARTPOLE: return eval<CartPoleModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_UNICYCLE: return eval<UnicycleModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_QUADROTOR: return eval<QuadrotorModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_QUADROTOR_EULER12: return eval<Quad12Model>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_MANIPULATOR: return eval<ManipulatorModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_MANIPULATOR7: return eval<Manip7Model>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_BICYCLE: return eval<BicycleModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_HCW: return eval<HCWModel>(integrator, dt, params, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    case CDDP_HIP_MODEL_CAR: {   // a discrete plant: its step needs the timestep, which travels as params[1] (as in the device descriptor)
      double pc[CDDP_HIP_MAX_MODEL_PARAMS];
      for (int i = 0; i < CDDP_HIP_MAX_MODEL_PARAMS; ++i) pc[i] = params[i];
      pc[1] = dt;
      return eval<CarModel>(integrator, dt, pc, nx, nu, x, u, x_next, fx, fu, fxx, fuu, fux, err);
    }
    default: err = "no host evaluation for this model id (LTI plants are evaluated by the caller: x+ = A x + B u)"; return -2;
  }
}

