/* A user plug-in written in C against include/cddp_hip.h (bench.py's plug-in line, tests/test_host_plugins.py): the pendulum of
 * examples/cddp_pendulum.cpp as DynamicalSystem / Objective / Constraint CALLBACKS -- continuous dynamics with the +sin convention of
 * src/dynamics_model/pendulum.cpp:29-66 stepped by Euler, analytic Jacobians, the quadratic objective of objective.cpp:80-154 (Q dt, R dt,
 * Q_f, reference 0), one control box.  Thread-safe (no state): cddp_hip_plugin_set_host_threads may fan the trajectories out.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o pendulum_plugin.so pendulum_plugin.c -lm */
#include <math.h>
typedef struct { double dt, length, mass, damping, gravity, Qf, R, umax; } pend_params;

void pend_dynamics(void *user, const double *x, const double *u, double time, double *xn) {
  const pend_params *p = (const pend_params *)user; (void)time;
  const double inertia = p->mass * p->length * p->length;
  const double f0 = x[1], f1 = (u[0] - p->damping * x[1] + p->mass * p->gravity * p->length * sin(x[0])) / inertia;
  xn[0] = x[0] + p->dt * f0; xn[1] = x[1] + p->dt * f1;
}
void pend_jacobians(void *user, const double *x, const double *u, double time, double *fx, double *fu) {
  const pend_params *p = (const pend_params *)user; (void)u; (void)time;
  fx[0] = 0.0; fx[1] = 1.0; fx[2] = (p->gravity / p->length) * cos(x[0]); fx[3] = -p->damping / (p->mass * p->length * p->length);
  fu[0] = 0.0; fu[1] = 1.0 / (p->mass * p->length * p->length);
}
double pend_running_cost(void *user, const double *x, const double *u, int index) {
  const pend_params *p = (const pend_params *)user; (void)x; (void)index;
  return (x[0] * (0.0 * x[0]) + x[1] * (0.0 * x[1])) + u[0] * ((p->R * p->dt) * u[0]);
}
double pend_terminal_cost(void *user, const double *x) {
  const pend_params *p = (const pend_params *)user;
  return x[0] * (p->Qf * x[0]) + x[1] * (p->Qf * x[1]);
}
void pend_running_cost_derivatives(void *user, const double *x, const double *u, int index, double *lx, double *lu, double *lxx, double *luu, double *lux) {
  const pend_params *p = (const pend_params *)user; (void)x; (void)index;
  lx[0] = 0.0; lx[1] = 0.0; lu[0] = 2.0 * ((p->R * p->dt) * u[0]);
  lxx[0] = lxx[1] = lxx[2] = lxx[3] = 0.0; luu[0] = 2.0 * (p->R * p->dt); lux[0] = lux[1] = 0.0;
}
void pend_terminal_cost_derivatives(void *user, const double *x, double *lx, double *lxx) {
  const pend_params *p = (const pend_params *)user;
  lx[0] = 2.0 * (p->Qf * x[0]); lx[1] = 2.0 * (p->Qf * x[1]);
  lxx[0] = 2.0 * p->Qf; lxx[1] = 0.0; lxx[2] = 0.0; lxx[3] = 2.0 * p->Qf;
}
/* BoxConstraint<Control>: evaluate = [-u; u], upper bound = [-lower; upper] (constraint.hpp:144-251); g = evaluate - upper */
void pend_constraints(void *user, const double *x, const double *u, int index, double *g, double *gx, double *gu) {
  const pend_params *p = (const pend_params *)user; (void)x; (void)index;
  g[0] = -u[0] - p->umax; g[1] = u[0] - p->umax;
  if (gx) { gx[0] = gx[1] = gx[2] = gx[3] = 0.0; }
  if (gu) { gu[0] = -1.0; gu[1] = 1.0; }
}
