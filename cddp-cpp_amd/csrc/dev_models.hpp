// Device-side plants: continuous dynamics f(x,u) and continuous-time Jacobians f_x, f_u
// (the DynamicalSystem plugin surface, reference include/cddp-cpp/cddp_core/dynamical_system.hpp
// and src/dynamics_model/*.cpp), enumerated by cddp_hip_model id.
//
// Derivative source per model:
//   Pendulum / Unicycle : the reference's analytic Jacobians (pendulum.cpp:44-66, unicycle.cpp:44-66)
//   CartPole            : hand-derived exact derivatives of the autodiff expression
//                         (cartpole.cpp:69-103, which includes the damping term)
//   Quadrotor           : forward-mode duals through the quaternion normalisation, i.e. what
//                         autodiff::jacobian does on quadrotor.cpp:166-219
//   Manipulator         : central finite differences, h = 2e-5 (manipulator.cpp:53-70, helper.hpp:95-118)
//   LTI                 : (A - I)/dt, B/dt (lti_system.cpp:78-92)
#pragma once
#include "dev_linalg.hpp"
#include "dev_trig.hpp"
#include "../../include/cddp_hip.h"

namespace cddp_dev {

// ---- forward-mode dual number with NP compile-time seeds (register resident) ------------
template <int NP>
struct DualN {
  double v;
  double d[NP];
  DEV DualN() {}
  DEV DualN(double x) : v(x) {
#pragma unroll
    for (int i = 0; i < NP; ++i) d[i] = 0.0;
  }
};
template <int NP> DEV DualN<NP> operator+(const DualN<NP> &a, const DualN<NP> &b) { DualN<NP> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int NP> DEV DualN<NP> operator-(const DualN<NP> &a, const DualN<NP> &b) { DualN<NP> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int NP> DEV DualN<NP> operator-(const DualN<NP> &a) { DualN<NP> r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = -a.d[i]; return r; }
template <int NP> DEV DualN<NP> operator*(const DualN<NP> &a, const DualN<NP> &b) { DualN<NP> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int NP> DEV DualN<NP> operator/(const DualN<NP> &a, const DualN<NP> &b) { DualN<NP> r; r.v = a.v / b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
template <int NP> DEV DualN<NP> dsin(const DualN<NP> &a) { DualN<NP> r; double s, c; plant_sincos(a.v, &s, &c); r.v = s;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = c * a.d[i]; return r; }
template <int NP> DEV DualN<NP> dcos(const DualN<NP> &a) { DualN<NP> r; double s, c; plant_sincos(a.v, &s, &c); r.v = c; double ms = -s;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = ms * a.d[i]; return r; }
template <int NP> DEV DualN<NP> dsqrt(const DualN<NP> &a) { DualN<NP> r; r.v = sqrt(a.v); double g = 0.5 / r.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = g * a.d[i]; return r; }
template <int NP> DEV DualN<NP> dasin(const DualN<NP> &a) { DualN<NP> r; r.v = plant_asin(a.v); double g = 1.0 / sqrt(1.0 - a.v * a.v);
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = g * a.d[i]; return r; }
template <int NP> DEV DualN<NP> dtan(const DualN<NP> &a) { DualN<NP> r; r.v = plant_tan(a.v); double g = 1.0 + r.v * r.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = g * a.d[i]; return r; }
DEV double dasin(double a) { return plant_asin(a); }
DEV double dtan(double a) { return plant_tan(a); }
DEV double dsin(double a) { return plant_sin(a); }
DEV double dcos(double a) { return plant_cos(a); }
DEV double dsqrt(double a) { return sqrt(a); }
DEV double dval(double a) { return a; }
template <int NP> DEV double dval(const DualN<NP> &a) { return a.v; }
// sin / cos of N scalars of type S (double or a dual) from ONE batched evaluation of the values (dev_trig.hpp: the
// independent angles of a plant share a basic block, so their chains interleave).  Used by the two large synthetic plants
// (12 / 52 evaluations per RK4 step); the reference's own plants keep the device libm -- there the rollout is not bound
// by it (no change measured at C2 / C3) and the libm's results left every strict parity case bit-stable.
DEV double lift_sin(double, double s, double) { return s; }
DEV double lift_cos(double, double, double c) { return c; }
template <int NP> DEV DualN<NP> lift_sin(const DualN<NP> &a, double s, double c) { DualN<NP> r; r.v = s;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = c * a.d[i]; return r; }
template <int NP> DEV DualN<NP> lift_cos(const DualN<NP> &a, double s, double c) { DualN<NP> r; r.v = c; const double ms = -s;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = ms * a.d[i]; return r; }
template <int N, class S> DEV void trig_n(const S *a, S *sn, S *cs) {
  double av[N], sv[N], cv[N];
#pragma unroll
  for (int i = 0; i < N; ++i) av[i] = dval(a[i]);
  sincos_n<N>(av, sv, cv);
#pragma unroll
  for (int i = 0; i < N; ++i) { sn[i] = lift_sin(a[i], sv[i], cv[i]); cs[i] = lift_cos(a[i], sv[i], cv[i]); }
}

// Second-order forward mode (the reference's autodiff::dual2nd, dynamical_system.cpp:137-217): value, gradient and Hessian
// w.r.t. NP seeded variables z = [x, u].  Only for the plants whose Hessians the reference takes from autodiff (CartPole)
// and only evaluated when options.use_ilqr == 0.
template <int NP>
struct Dual2N {
  double v, d[NP], h[NP * NP];
  DEV Dual2N() {}
  DEV Dual2N(double x) : v(x) {
    for (int i = 0; i < NP; ++i) d[i] = 0.0;
    for (int i = 0; i < NP * NP; ++i) h[i] = 0.0;
  }
};
template <int NP> DEV Dual2N<NP> d2_unary(const Dual2N<NP> &a, double val, double p1, double p2) {
  Dual2N<NP> r; r.v = val;
  for (int i = 0; i < NP; ++i) { r.d[i] = p1 * a.d[i]; for (int j = 0; j < NP; ++j) r.h[i * NP + j] = p2 * a.d[i] * a.d[j] + p1 * a.h[i * NP + j]; }
  return r;
}
template <int NP> DEV Dual2N<NP> operator+(const Dual2N<NP> &a, const Dual2N<NP> &b) { Dual2N<NP> r; r.v = a.v + b.v;
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i];
  for (int i = 0; i < NP * NP; ++i) r.h[i] = a.h[i] + b.h[i]; return r; }
template <int NP> DEV Dual2N<NP> operator-(const Dual2N<NP> &a, const Dual2N<NP> &b) { Dual2N<NP> r; r.v = a.v - b.v;
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i];
  for (int i = 0; i < NP * NP; ++i) r.h[i] = a.h[i] - b.h[i]; return r; }
template <int NP> DEV Dual2N<NP> operator-(const Dual2N<NP> &a) { Dual2N<NP> r; r.v = -a.v;
  for (int i = 0; i < NP; ++i) r.d[i] = -a.d[i];
  for (int i = 0; i < NP * NP; ++i) r.h[i] = -a.h[i]; return r; }
template <int NP> DEV Dual2N<NP> operator*(const Dual2N<NP> &a, const Dual2N<NP> &b) { Dual2N<NP> r; r.v = a.v * b.v;
  for (int i = 0; i < NP; ++i) {
    r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    for (int j = 0; j < NP; ++j) r.h[i * NP + j] = a.h[i * NP + j] * b.v + a.d[i] * b.d[j] + a.d[j] * b.d[i] + a.v * b.h[i * NP + j];
  }
  return r; }
template <int NP> DEV Dual2N<NP> operator/(const Dual2N<NP> &a, const Dual2N<NP> &b) {
  const double inv = 1.0 / b.v;
  return a * d2_unary<NP>(b, inv, -inv * inv, 2.0 * inv * inv * inv); }
template <int NP> DEV Dual2N<NP> dsin(const Dual2N<NP> &a) { double s, c; plant_sincos(a.v, &s, &c); return d2_unary<NP>(a, s, c, -s); }
template <int NP> DEV Dual2N<NP> dcos(const Dual2N<NP> &a) { double s, c; plant_sincos(a.v, &s, &c); return d2_unary<NP>(a, c, -s, -c); }
template <int NP> DEV Dual2N<NP> dsqrt(const Dual2N<NP> &a) { const double r = sqrt(a.v); return d2_unary<NP>(a, r, 0.5 / r, -0.25 / (a.v * r)); }
template <int NP> DEV Dual2N<NP> dasin(const Dual2N<NP> &a) { const double w = 1.0 - a.v * a.v, r = sqrt(w); return d2_unary<NP>(a, plant_asin(a.v), 1.0 / r, a.v / (w * r)); }
template <int NP> DEV Dual2N<NP> dtan(const Dual2N<NP> &a) { const double t = plant_tan(a.v), g = 1.0 + t * t; return d2_unary<NP>(a, t, g, 2.0 * t * g); }
template <int NP> DEV double dval(const Dual2N<NP> &a) { return a.v; }
template <int NP> DEV Dual2N<NP> lift_sin(const Dual2N<NP> &a, double s, double c) { return d2_unary<NP>(a, s, c, -s); }   // trig_n on second-order duals
template <int NP> DEV Dual2N<NP> lift_cos(const Dual2N<NP> &a, double s, double c) { return d2_unary<NP>(a, c, -s, -c); }

// Hessian tensors by second-order duals of a templated functor F::template eval<S>(p, x, u, out): out_i's Hessian w.r.t. z = [x, u],
// split into the solver's three blocks (f_xx[i] nx x nx, f_uu[i] nu x nu, f_ux[i] nu x nx), each entry divided by `div`
template <class F, int NX, int NU>
DEV void ad_hessian(const double *p, const double *x, const double *u, double div, double *Fxx, double *Fuu, double *Fux) {
  typedef Dual2N<NX + NU> D;
  D xs[NX], us[NU], xd[NX];
  for (int i = 0; i < NX; ++i) { xs[i] = D(x[i]); xs[i].d[i] = 1.0; }
  for (int j = 0; j < NU; ++j) { us[j] = D(u[j]); us[j].d[NX + j] = 1.0; }
  F::template eval<D>(p, xs, us, xd);
  constexpr int NP = NX + NU;
  for (int i = 0; i < NX; ++i) {
    for (int a = 0; a < NX; ++a) for (int b = 0; b < NX; ++b) Fxx[(i * NX + a) * NX + b] = xd[i].h[a * NP + b] / div;
    for (int a = 0; a < NU; ++a) for (int b = 0; b < NU; ++b) Fuu[(i * NU + a) * NU + b] = xd[i].h[(NX + a) * NP + NX + b] / div;
    for (int a = 0; a < NU; ++a) for (int b = 0; b < NX; ++b) Fux[(i * NU + a) * NX + b] = xd[i].h[(NX + a) * NP + b] / div;
  }
}

// The same second-order terms WITHOUT materialising the tensors, for plants whose full dual2nd frame does not fit a GPU lane
// (quadrotor: 17 seeds = 307 doubles per dual number, a 180-KB private frame; 7-joint arm: 21 seeds).  The Hessian of every output
// is assembled from evaluations with 2 * BS seeds at a time: the seeded variables z = [x, u] are cut into blocks of BS, and for every
// unordered pair of blocks (I, J) the functor is evaluated once with the variables of I in seed slots 0 .. BS-1 and those of J in
// slots BS .. 2 BS - 1 (73 doubles per dual number at BS = 4).  Second-order forward mode is component-wise in the seed pair --
// entry (a, b) of a product or a unary lift reads only the operands' value, d[a], d[b] and h[a][b] (operator* / d2_unary above) --
// so every entry equals, bit for bit, the entry the full Dual2N<NX + NU> evaluation (ad_hessian, the host build) produces.
// Accumulated directly into the solver's sums in its own order (for i: Q[e] += w[i] * (dt * F[i][e]), i ascending per entry):
//   Qxx[a * NX + b] += w[i] (dt f_i,xx[a][b] / div),  Qux[a * NX + b] += w[i] (dt f_i,ux[a][b] / div),  Quu likewise.
template <class F, int NX, int NU, int BS>
DEV_NOINLINE void ad_tensor_terms_blocked(const double *p, const double *x, const double *u, const double *w, double dt, double div,
                                          double *Qxx, double *Qux, double *Quu) {
  constexpr int NP = NX + NU, NBK = (NP + BS - 1) / BS, NS = 2 * BS;
  typedef Dual2N<NS> D;
  for (int I = 0; I < NBK; ++I) {
    for (int J = I; J < NBK; ++J) {
      D xs[NX], us[NU], xd[NX];
      for (int v = 0; v < NP; ++v) {
        D z(v < NX ? x[v] : u[v - NX]);
        const int blk = v / BS, off = v - blk * BS;
        if (blk == I) z.d[off] = 1.0;
        else if (blk == J) z.d[BS + off] = 1.0;
        if (v < NX) xs[v] = z; else us[v - NX] = z;
      }
      F::template eval<D>(p, xs, us, xd);
      // entries (row variable va, column variable vb) this pair owns: va in I, vb in J, and (I != J) va in J, vb in I
      for (int pass = 0; pass < (I == J ? 1 : 2); ++pass) {
        const int RB = pass == 0 ? I : J, CB = pass == 0 ? J : I;
        for (int ra = 0; ra < BS; ++ra) {
          const int va = RB * BS + ra;
          if (va >= NP) break;
          const int sa = (RB == I) ? ra : BS + ra;
          for (int cb = 0; cb < BS; ++cb) {
            const int vb = CB * BS + cb;
            if (vb >= NP) break;
            const int sb = (CB == I) ? cb : BS + cb;
            double *dst;
            if (va < NX && vb < NX) dst = Qxx + va * NX + vb;
            else if (va >= NX && vb >= NX) dst = Quu + (va - NX) * NU + (vb - NX);
            else if (va >= NX && vb < NX) dst = Qux + (va - NX) * NX + vb;
            else continue;   // d2 f / dx du: the solver reads the (u, x) block only
            double acc = *dst;
            for (int i = 0; i < NX; ++i) acc = acc + w[i] * (dt * (xd[i].h[sa * NS + sb] / div));
            *dst = acc;
          }
        }
      }
    }
  }
}

// does the plant take its second-order terms from the blocked evaluation (a `HessDyn` functor typedef, device build only)?
template <class M, class = void> struct HessBlocked { static constexpr bool value = false; };
template <class M> struct HessBlocked<M, decltype((void)sizeof(typename M::HessDyn))> { static constexpr bool value = true; };

// Jacobian by forward-mode duals of a templated dynamics functor F::template eval<S>(p, x, u, xd)
template <class F, int NX, int NU>
DEV void ad_jacobian(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
  typedef DualN<NX + NU> D;
  D xs[NX], us[NU], xd[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) { xs[i] = D(x[i]); xs[i].d[i] = 1.0; }
#pragma unroll
  for (int j = 0; j < NU; ++j) { us[j] = D(u[j]); us[j].d[NX + j] = 1.0; }
  F::template eval<D>(p, xs, us, xd);
#pragma unroll
  for (int i = 0; i < NX; ++i) {
#pragma unroll
    for (int j = 0; j < NX; ++j) Fx[i * NX + j] = xd[i].d[j];
#pragma unroll
    for (int j = 0; j < NU; ++j) Fu[i * NU + j] = xd[i].d[NX + j];
  }
}

// ================================================================================ Pendulum
struct PendulumModel {   // pendulum.cpp:29-66; params: length, mass, damping, gravity
  static constexpr int ID = CDDP_HIP_MODEL_PENDULUM, NX = 2, NU = 1;
  static constexpr bool kDiscrete = false;
  DEV static void f(const double *p, const double *x, const double *u, double *xd) {
    const double length = p[0], mass = p[1], damping = p[2], gravity = p[3];
    const double inertia = mass * length * length;
    xd[0] = x[1];
    xd[1] = (u[0] - damping * x[1] + mass * gravity * length * dsin(x[0])) / inertia;
  }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    const double length = p[0], mass = p[1], damping = p[2], gravity = p[3];
    Fx[0] = 0.0; Fx[1] = 1.0;
    Fx[2] = (gravity / length) * dcos(x[0]);
    Fx[3] = -damping / (mass * length * length);
    Fu[0] = 0.0; Fu[1] = 1.0 / (mass * length * length);
  }
  // Hessian tensors f_xx[i] (NX x NX), f_uu[i] (NU x NU), f_ux[i] (NU x NX), i = output row: state Hessian analytic
  // (pendulum.cpp:68-78), control Hessian zero (:80-85), cross Hessian = autodiff of the -sin twin (:87-100) = zero
  static constexpr bool kHasHess = true;
  DEV static void hess(const double *p, const double *x, const double *, double *Fxx, double *Fuu, double *Fux) {
    for (int i = 0; i < NX * NX * NX; ++i) Fxx[i] = 0.0;
    for (int i = 0; i < NX * NU * NU; ++i) Fuu[i] = 0.0;
    for (int i = 0; i < NX * NU * NX; ++i) Fux[i] = 0.0;
    Fxx[1 * NX * NX + 0] = -(p[3] / p[0]) * dsin(x[0]);
  }
};

// ================================================================================ CartPole
struct CartPoleModel {   // cartpole.cpp:38-103; params: cart_mass, pole_mass, pole_length, gravity, damping
  static constexpr int ID = CDDP_HIP_MODEL_CARTPOLE, NX = 4, NU = 1;
  static constexpr bool kDiscrete = false;
  DEV static void f(const double *p, const double *x, const double *u, double *xd) {
    // double path (cartpole.cpp:38-67): NO damping term
    const double mc = p[0], mp = p[1], l = p[2], g = p[3];
    const double theta_dot = x[3], force = u[0];
    double s, c; plant_sincos(x[1], &s, &c);
    const double total_mass = mc + mp;
    const double den = mc + mp * s * s;
    xd[0] = x[2];
    xd[1] = theta_dot;
    xd[2] = (force + mp * s * (l * theta_dot * theta_dot + g * c)) / den;
    xd[3] = (-force * c - mp * l * theta_dot * theta_dot * c * s - total_mass * g * s) / (l * den);
  }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
#ifdef CDDP_TRIG_SHARED
    // parity build: forward-mode duals through the autodiff expression, operation for operation what autodiff::jacobian does
    // (and what the CPU checker restates) -- the hand-derived form below is the same derivative from a different expression
    // tree, i.e. equal up to the last bits only, and a non-converging solve amplifies last bits
    {
      typedef DualN<5> D;
      D xs[4], us[1], xd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { xs[i] = D(x[i]); xs[i].d[i] = 1.0; }
      us[0] = D(u[0]); us[0].d[4] = 1.0;
      f_ad<D>(p, xs, us, xd);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Fx[i * 4 + j] = xd[i].d[j];
        Fu[i] = xd[i].d[4];
      }
      return;
    }
#endif
    // exact derivatives of the autodiff expression (cartpole.cpp:69-103, WITH -damping*theta_dot)
    const double mc = p[0], mp = p[1], l = p[2], g = p[3], b = p[4];
    const double w = x[3], F = u[0];
    double s, c; plant_sincos(x[1], &s, &c);
    const double M = mc + mp;
    const double den = mc + mp * s * s;
    const double dden = 2.0 * mp * s * c;
    const double inner = l * w * w + g * c;
    const double num3 = F + mp * s * inner;
    const double dnum3 = mp * c * inner - mp * s * g * s;
    const double num4 = -F * c - mp * l * w * w * c * s - M * g * s - b * w;
    const double dnum4 = F * s - mp * l * w * w * (c * c - s * s) - M * g * c;
    const double den2 = den * den;
#pragma unroll
    for (int i = 0; i < 16; ++i) Fx[i] = 0.0;
    Fx[0 * 4 + 2] = 1.0;
    Fx[1 * 4 + 3] = 1.0;
    Fx[2 * 4 + 1] = (dnum3 * den - num3 * dden) / den2;
    Fx[2 * 4 + 3] = (mp * s * (2.0 * l * w)) / den;
    Fx[3 * 4 + 1] = (dnum4 * den - num4 * dden) / (l * den2);
    Fx[3 * 4 + 3] = (-2.0 * mp * l * w * c * s - b) / (l * den);
    Fu[0] = 0.0; Fu[1] = 0.0;
    Fu[2] = 1.0 / den;
    Fu[3] = -c / (l * den);
  }
  // the autodiff expression (cartpole.cpp:69-103, with the damping term) on any scalar type
  template <class S>
  DEV static void f_ad(const double *p, const S *x, const S *u, S *xd) {
    const double mc = p[0], mp = p[1], l = p[2], g = p[3], b = p[4];
    const S theta = x[1], x_dot = x[2], theta_dot = x[3], force = u[0];
    const S sin_theta = dsin(theta), cos_theta = dcos(theta);
    const double total_mass = mc + mp;
    const S den = S(mc) + S(mp) * sin_theta * sin_theta;
    xd[0] = x_dot;
    xd[1] = theta_dot;
    xd[2] = (force + S(mp) * sin_theta * (S(l) * theta_dot * theta_dot + S(g) * cos_theta)) / den;
    xd[3] = (-force * cos_theta - S(mp) * S(l) * theta_dot * theta_dot * cos_theta * sin_theta - S(total_mass) * S(g) * sin_theta - S(b) * theta_dot) / (S(l) * den);
  }
  // Hessians: the DynamicalSystem defaults (dual2nd through the autodiff path, cartpole.cpp:191-199)
  static constexpr bool kHasHess = true;
  DEV static void hess(const double *p, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) {
    typedef Dual2N<5> D;
    D xs[4], us[1], xd[4];
    for (int i = 0; i < 4; ++i) { xs[i] = D(x[i]); xs[i].d[i] = 1.0; }
    us[0] = D(u[0]); us[0].d[4] = 1.0;
    f_ad<D>(p, xs, us, xd);
    for (int i = 0; i < 4; ++i) {
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) Fxx[i * 16 + a * 4 + b] = xd[i].h[a * 5 + b];
      Fuu[i] = xd[i].h[4 * 5 + 4];
      for (int b = 0; b < 4; ++b) Fux[i * 4 + b] = xd[i].h[4 * 5 + b];
    }
  }
};

// ================================================================================ Unicycle
struct UnicycleModel {   // unicycle.cpp:28-66
  static constexpr int ID = CDDP_HIP_MODEL_UNICYCLE, NX = 3, NU = 2;
  static constexpr bool kDiscrete = false;
  DEV static void f(const double *, const double *x, const double *u, double *xd) {
    double s, c; plant_sincos(x[2], &s, &c);
    xd[0] = u[0] * c; xd[1] = u[0] * s; xd[2] = u[1];
  }
  DEV static void jac(const double *, const double *x, const double *u, double *Fx, double *Fu) {
    double s, c; plant_sincos(x[2], &s, &c);
#pragma unroll
    for (int i = 0; i < 9; ++i) Fx[i] = 0.0;
    Fx[0 * 3 + 2] = -u[0] * s;
    Fx[1 * 3 + 2] = u[0] * c;
    Fu[0] = c; Fu[1] = 0.0; Fu[2] = s; Fu[3] = 0.0; Fu[4] = 0.0; Fu[5] = 1.0;
  }
  // state Hessian analytic (unicycle.cpp:68-80), control Hessian zero (:82-89), cross Hessian = the autodiff default on
  // getContinuousDynamicsAutodiff (:91-107): d2(v cos th)/dv dth = -sin th, d2(v sin th)/dv dth = cos th
  static constexpr bool kHasHess = true;
  DEV static void hess(const double *, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) {
    double s, c; plant_sincos(x[2], &s, &c);
    for (int i = 0; i < NX * NX * NX; ++i) Fxx[i] = 0.0;
    for (int i = 0; i < NX * NU * NU; ++i) Fuu[i] = 0.0;
    for (int i = 0; i < NX * NU * NX; ++i) Fux[i] = 0.0;
    Fxx[0 * 9 + 2 * 3 + 2] = -u[0] * c;
    Fxx[1 * 9 + 2 * 3 + 2] = -u[0] * s;
    Fux[0 * 6 + 0 * 3 + 2] = -s;
    Fux[1 * 6 + 0 * 3 + 2] = c;
  }
};

// ================================================================================ LTI
// Discrete x+ = A x + B u (lti_system.cpp:71-76); A at p[0 .. NX*NX), B after it.
template <int NX_, int NU_>
struct LTIModel {
  static constexpr int ID = CDDP_HIP_MODEL_LTI, NX = NX_, NU = NU_;
  static constexpr bool kDiscrete = true;
  static constexpr bool kHasHess = true;   // lti_system.cpp:94-115: zero
  DEV static void hess(const double *, const double *, const double *, double *Fxx, double *Fuu, double *Fux) {
    for (int i = 0; i < NX * NX * NX; ++i) Fxx[i] = 0.0;
    for (int i = 0; i < NX * NU * NU; ++i) Fuu[i] = 0.0;
    for (int i = 0; i < NX * NU * NX; ++i) Fux[i] = 0.0;
  }
  DEV static void step(const double *p, const double *x, const double *u, double *xn) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = 0.0, t = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += p[i * NX + j] * x[j];
#pragma unroll
      for (int j = 0; j < NU; ++j) t += p[NX * NX + i * NU + j] * u[j];
      xn[i] = s + t;
    }
  }
  DEV static void f(const double *, const double *, const double *, double *) {}
  // continuous-equivalent Jacobians; dt arrives in p[NX*NX + NX*NU]
  DEV static void jac(const double *p, const double *, const double *, double *Fx, double *Fu) {
    const double dt = p[NX * NX + NX * NU];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) Fx[i * NX + j] = (p[i * NX + j] - (i == j ? 1.0 : 0.0)) / dt;
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Fu[i] = p[NX * NX + i] / dt;
  }
};

// ================================================================================ Bicycle (bicycle.cpp)
// Kinematic bicycle, state [x, y, theta, v], control [a, delta]; params: wheelbase.  Analytic Jacobians (:68-111) and state /
// control Hessians (:113-156); the cross Hessian is the base class's autodiff default (dynamical_system.cpp:190-217).
struct BicycleDyn {
  template <class S>
  DEV static void eval(const double *p, const S *x, const S *u, S *xd) {
    const S theta = x[2], v = x[3];
    xd[0] = v * dcos(theta);
    xd[1] = v * dsin(theta);
    xd[2] = (v / S(p[0])) * dtan(u[1]);
    xd[3] = u[0];
  }
};
struct BicycleModel {
  static constexpr int ID = CDDP_HIP_MODEL_BICYCLE, NX = 4, NU = 2;
  static constexpr bool kDiscrete = false;
  static constexpr bool kHasHess = true;
  DEV static void f(const double *p, const double *x, const double *u, double *xd) { BicycleDyn::eval<double>(p, x, u, xd); }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    double s, c; plant_sincos(x[2], &s, &c);
    const double v = x[3], L = p[0], cd = plant_cos(u[1]);
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Fx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Fu[i] = 0.0;
    Fx[0 * NX + 2] = -v * s; Fx[0 * NX + 3] = c;
    Fx[1 * NX + 2] = v * c;  Fx[1 * NX + 3] = s;
    Fx[2 * NX + 3] = plant_tan(u[1]) / L;
    Fu[3 * NU + 0] = 1.0;
    Fu[2 * NU + 1] = v / (L * (cd * cd));
  }
  DEV static void hess(const double *p, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) {
    double sxx[NX * NX * NX], suu[NX * NU * NU];   // the autodiff blocks the analytic overrides replace
    ad_hessian<BicycleDyn, NX, NU>(p, x, u, 1.0, sxx, suu, Fux);
    double s, c; plant_sincos(x[2], &s, &c);
    const double v = x[3], L = p[0], cd = plant_cos(u[1]);
    for (int i = 0; i < NX * NX * NX; ++i) Fxx[i] = 0.0;
    for (int i = 0; i < NX * NU * NU; ++i) Fuu[i] = 0.0;
    Fxx[(0 * NX + 2) * NX + 2] = -v * c; Fxx[(0 * NX + 2) * NX + 3] = -s; Fxx[(0 * NX + 3) * NX + 2] = -s;
    Fxx[(1 * NX + 2) * NX + 2] = -v * s; Fxx[(1 * NX + 2) * NX + 3] = c;  Fxx[(1 * NX + 3) * NX + 2] = c;
    Fuu[(2 * NU + 1) * NU + 1] = 2.0 * v * plant_sin(u[1]) / (L * (cd * cd * cd));
  }
};

// ================================================================================ HCW (spacecraft_linear.cpp)
// Hill-Clohessy-Wiltshire equations of relative orbital motion: state [x, y, z, vx, vy, vz] (radial, along-track, cross-track), control
// [Fx, Fy, Fz]; params: mean motion n, mass.  Linear and time-invariant in continuous time: constant Jacobians (:56-83), zero Hessians
// (:85-120).
struct HCWModel {
  static constexpr int ID = CDDP_HIP_MODEL_HCW, NX = 6, NU = 3;
  static constexpr bool kDiscrete = false;
  static constexpr bool kHasHess = true;
  DEV static void f(const double *p, const double *x, const double *u, double *xd) {
    const double n = p[0], n2 = n * n, mass = p[1];
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
    xd[3] = 2.0 * n * x[4] + 3.0 * n2 * x[0] + u[0] / mass;
    xd[4] = -2.0 * n * x[3] + u[1] / mass;
    xd[5] = -n2 * x[2] + u[2] / mass;
  }
  DEV static void jac(const double *p, const double *, const double *, double *Fx, double *Fu) {
    const double n = p[0], n2 = n * n, mass = p[1];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Fx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Fu[i] = 0.0;
    Fx[0 * NX + 3] = 1.0; Fx[1 * NX + 4] = 1.0; Fx[2 * NX + 5] = 1.0;
    Fx[3 * NX + 0] = 3.0 * n2; Fx[3 * NX + 4] = 2.0 * n;
    Fx[4 * NX + 3] = -2.0 * n;
    Fx[5 * NX + 2] = -n2;
    Fu[3 * NU + 0] = 1.0 / mass; Fu[4 * NU + 1] = 1.0 / mass; Fu[5 * NU + 2] = 1.0 / mass;
  }
  DEV static void hess(const double *, const double *, const double *, double *Fxx, double *Fuu, double *Fux) {
    for (int i = 0; i < NX * NX * NX; ++i) Fxx[i] = 0.0;
    for (int i = 0; i < NX * NU * NU; ++i) Fuu[i] = 0.0;
    for (int i = 0; i < NX * NU * NX; ++i) Fux[i] = 0.0;
  }
};

// ================================================================================ Car (car.cpp)
// A DISCRETE plant: getDiscreteDynamics is overridden (:24-60) and everything else differentiates it -- Jacobians are the autodiff
// gradient of the discrete map with J.diagonal() -= 1 and J /= timestep (:62-111), Hessians its autodiff Hessian / timestep
// (:113-161; the cross Hessian through the base default on (x+ - x) / timestep is the same block).  State [x, y, theta, v],
// control [steering delta, acceleration a]; params: wheelbase, p[1] = timestep (filled in by the library).
struct CarDyn {
  template <class S, bool kClamps>
  DEV static void next(const double *p, const S *x, const S *u, S *xn) {
    const double d = p[0], h = p[1];
    const S theta = x[2], v = x[3], delta = u[0], a = u[1];
    const S cos_theta = dcos(theta), sin_theta = dsin(theta);
    const S f = S(h) * v;
    const S sin_delta = dsin(delta);
    const S f_sin_delta = f * sin_delta;
    S inside = S(d * d) - f_sin_delta * f_sin_delta;
    if (kClamps && dval(inside) < 0.0) inside = S(0.0);            // autodiff path only (:183-186)
    const S b = S(d) + f * dcos(delta) - dsqrt(inside);
    S asin_arg = sin_delta * f / S(d);
    if (kClamps && fabs(dval(asin_arg)) > 1.0) asin_arg = S(dval(asin_arg) > 0.0 ? 1.0 : -1.0);   // :196-199
    const S dtheta = dasin(asin_arg);
    xn[0] = x[0] + b * cos_theta;
    xn[1] = x[1] + b * sin_theta;
    xn[2] = x[2] + dtheta;
    xn[3] = x[3] + S(h) * a;
  }
  template <class S>
  DEV static void eval(const double *p, const S *x, const S *u, S *xn) { next<S, true>(p, x, u, xn); }
};
struct CarModel {
  static constexpr int ID = CDDP_HIP_MODEL_CAR, NX = 4, NU = 2;
  static constexpr bool kDiscrete = true;
  static constexpr bool kHasHess = true;
  DEV static void step(const double *p, const double *x, const double *u, double *xn) { CarDyn::next<double, false>(p, x, u, xn); }
  DEV static void f(const double *, const double *, const double *, double *) {}
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    ad_jacobian<CarDyn, NX, NU>(p, x, u, Fx, Fu);
    const double h = p[1];
#pragma unroll
    for (int i = 0; i < NX; ++i) Fx[i * NX + i] -= 1.0;
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Fx[i] = Fx[i] / h;
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Fu[i] = Fu[i] / h;
  }
  DEV static void hess(const double *p, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) {
    ad_hessian<CarDyn, NX, NU>(p, x, u, p[1], Fxx, Fuu, Fux);
  }
};

// ================================================================================ Quadrotor (nx=13)
struct QuadrotorDyn {   // quadrotor.cpp:33-104 == :166-219; params: mass, arm, Ixx, Iyy, Izz, gravity
  template <class S>
  DEV static void eval(const double *p, const S *x, const S *u, S *xd) {
    const double mass = p[0], arm = p[1], Ixx = p[2], Iyy = p[3], Izz = p[4], grav = p[5];
    xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
    S qw = x[3], qx = x[4], qy = x[5], qz = x[6];
    S norm = dsqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    if (dval(norm) > 1e-6) { qw = qw / norm; qx = qx / norm; qy = qy / norm; qz = qz / norm; }
    else { qw = S(1.0); qx = S(0.0); qy = S(0.0); qz = S(0.0); }
    const S ox = x[10], oy = x[11], oz = x[12];
    xd[3] = S(-0.5) * (qx * ox + qy * oy + qz * oz);
    xd[4] = S(0.5) * (qw * ox + qy * oz - qz * oy);
    xd[5] = S(0.5) * (qw * oy - qx * oz + qz * ox);
    xd[6] = S(0.5) * (qw * oz + qx * oy - qy * ox);
    const S f1 = u[0], f2 = u[1], f3 = u[2], f4 = u[3];
    const S thrust = f1 + f2 + f3 + f4;
    const S tau_x = S(arm) * (f1 - f3);
    const S tau_y = S(arm) * (f2 - f4);
    const S tau_z = S(0.1) * (f1 - f2 + f3 - f4);
    const S R02 = S(2.0) * (qx * qz + qy * qw);
    const S R12 = S(2.0) * (qy * qz - qx * qw);
    const S R22 = S(1.0) - S(2.0) * (qx * qx + qy * qy);
    const double invm = 1.0 / mass;
    xd[7] = S(invm) * (R02 * thrust);
    xd[8] = S(invm) * (R12 * thrust);
    xd[9] = S(invm) * (R22 * thrust) - S(grav);
    const double c00 = Iyy * Izz, c11 = Ixx * Izz, c22 = Ixx * Iyy;
    const double det = c00 * Ixx;
    const double invdet = 1.0 / det;
    const double i00 = c00 * invdet, i11 = c11 * invdet, i22 = c22 * invdet;
    const S Iox = S(Ixx) * ox, Ioy = S(Iyy) * oy, Ioz = S(Izz) * oz;
    const S cx = oy * Ioz - oz * Ioy;
    const S cy = oz * Iox - ox * Ioz;
    const S cz = ox * Ioy - oy * Iox;
    xd[10] = S(i00) * (tau_x - cx);
    xd[11] = S(i11) * (tau_y - cy);
    xd[12] = S(i22) * (tau_z - cz);
  }
};
struct QuadrotorModel {
  static constexpr int ID = CDDP_HIP_MODEL_QUADROTOR, NX = 13, NU = 4;
  static constexpr bool kDiscrete = false;
  // quadrotor.cpp:224-278: dual2nd through the normalised-quaternion dynamics (17 seeds: 307 doubles per dual number).  On the
  // device that is a 180-KB private frame per lane -- beyond the 128-KB limit -- so the tensors exist in the HOST build of this file
  // only (host_models.cpp -> cddp_hip_model_eval -> the plug-in solve), and the device-resident solve refuses use_ilqr = 0 for this
  // plant.  Little is lost: the reference's own getCrossHessian returns nx x nu matrices where its solvers add nu x nx blocks (an
  // Eigen size mismatch), i.e. full DDP on the quadrotor is not defined behaviour there; here the cross block has the shape the
  // solver's formula needs.
#ifdef CDDP_HOST_MODELS
  static constexpr bool kHasHess = true;
  static void hess(const double *p, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) {
    ad_hessian<QuadrotorDyn, NX, NU>(p, x, u, 1.0, Fxx, Fuu, Fux);
  }
  static constexpr bool kHessBlocked = false;
#else
  static constexpr bool kHasHess = false;
  // device (round 4): the tensors are never materialised -- their contraction with the value gradient is assembled from 8-seed
  // evaluations (ad_tensor_terms_blocked, bitwise the full 17-seed evaluation of the host build)
  static constexpr bool kHessBlocked = true;
  typedef QuadrotorDyn HessDyn;
  static constexpr double kHessDiv = 1.0;
#endif
  DEV static void f(const double *p, const double *x, const double *u, double *xd) { QuadrotorDyn::eval<double>(p, x, u, xd); }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    ad_jacobian<QuadrotorDyn, NX, NU>(p, x, u, Fx, Fu);
  }
};

// ================================================================================ SYNTHETIC quadrotor, Euler-ZYX, nx=12
struct Quad12Dyn {
  template <class S>
  DEV static void eval(const double *p, const S *x, const S *u, S *xd) {
    const double mass = p[0], arm = p[1], Ixx = p[2], Iyy = p[3], Izz = p[4], grav = p[5];
    const S phi = x[6], th = x[7], psi = x[8];
    const S ox = x[9], oy = x[10], oz = x[11];
    const S ang[3] = {phi, th, psi}; S sn[3], cs[3];
    trig_n<3, S>(ang, sn, cs);
    const S sph = sn[0], cph = cs[0], sth = sn[1], cth = cs[1], sps = sn[2], cps = cs[2];
    const S thrust = u[0] + u[1] + u[2] + u[3];
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
    const double invm = 1.0 / mass;
    xd[3] = S(invm) * ((cph * sth * cps + sph * sps) * thrust);
    xd[4] = S(invm) * ((cph * sth * sps - sph * cps) * thrust);
    xd[5] = S(invm) * ((cph * cth) * thrust) - S(grav);
    const S tth = sth / cth;
    xd[6] = ox + sph * tth * oy + cph * tth * oz;
    xd[7] = cph * oy - sph * oz;
    xd[8] = (sph * oy + cph * oz) / cth;
    const S tau_x = S(arm) * (u[0] - u[2]);
    const S tau_y = S(arm) * (u[1] - u[3]);
    const S tau_z = S(0.1) * (u[0] - u[1] + u[2] - u[3]);
    xd[9] = (tau_x - (S(Izz) - S(Iyy)) * oy * oz) / S(Ixx);
    xd[10] = (tau_y - (S(Ixx) - S(Izz)) * oz * ox) / S(Iyy);
    xd[11] = (tau_z - (S(Iyy) - S(Ixx)) * ox * oy) / S(Izz);
  }
};
struct Quad12Model {
  static constexpr int ID = CDDP_HIP_MODEL_QUADROTOR_EULER12, NX = 12, NU = 4;
  static constexpr bool kDiscrete = false;
  // full DDP (round 4): second-order duals through the plant's own expression, as the reference's base class does for a plant
  // without overrides (dynamical_system.cpp:137-217); host build: the whole tensors, device: blocked contraction
#ifdef CDDP_HOST_MODELS
  static constexpr bool kHasHess = true;
  static constexpr bool kHessBlocked = false;
  static void hess(const double *p, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) { ad_hessian<Quad12Dyn, NX, NU>(p, x, u, 1.0, Fxx, Fuu, Fux); }
#else
  static constexpr bool kHasHess = false;
  static constexpr bool kHessBlocked = true;
  typedef Quad12Dyn HessDyn;
  static constexpr double kHessDiv = 1.0;
#endif
  DEV static void f(const double *p, const double *x, const double *u, double *xd) { Quad12Dyn::eval<double>(p, x, u, xd); }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    ad_jacobian<Quad12Dyn, NX, NU>(p, x, u, Fx, Fu);
  }
};

// ================================================================================ Manipulator (3-DOF)
struct ManipulatorModel {   // manipulator.cpp:29-70,174-208
  static constexpr int ID = CDDP_HIP_MODEL_MANIPULATOR, NX = 6, NU = 3;
  static constexpr bool kDiscrete = false;
  // manipulator.cpp:72-86: the state and control Hessians are ZERO overrides; the cross Hessian is the base class's autodiff default
  // on ddq = M(q)^-1 (tau - G(q)) (dynamical_system.cpp:190-217), i.e. d2 ddq_i / dtau_j dq_k = d(M^-1)_ij / dq_k, evaluated here in
  // closed form: -(M^-1 (dM/dq_k) M^-1)_ij, k = 1, 2 (M does not depend on q_0).
  static constexpr bool kHasHess = true;
  DEV static void hess(const double *, const double *x, const double *, double *Fxx, double *Fuu, double *Fux) {
    const double la = 1.0, lb = 0.2, lc = 1.0, m1 = 1.0, m2 = 1.0, m3 = 0.5;
    for (int i = 0; i < NX * NX * NX; ++i) Fxx[i] = 0.0;
    for (int i = 0; i < NX * NU * NU; ++i) Fuu[i] = 0.0;
    for (int i = 0; i < NX * NU * NX; ++i) Fux[i] = 0.0;
    double s1, c1, s2, c2, s12, c12;
    plant_sincos(x[1], &s1, &c1); plant_sincos(x[2], &s2, &c2); plant_sincos(x[1] + x[2], &s12, &c12);
    double M[9], Minv[9];
    M[0] = (m1 + m2 + m3) * (la * la); M[4] = (m2 + m3) * (lb * lb); M[8] = m3 * (lc * lc);
    M[1] = M[3] = (m2 + m3) * la * lb * c1; M[5] = M[7] = m3 * lb * lc * c2; M[2] = M[6] = m3 * la * lc * c12;
    inverse_pplu<3>(M, Minv);
    for (int k = 1; k <= 2; ++k) {
      double dM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      dM[2] = dM[6] = -(m3 * la * lc) * s12;
      if (k == 1) dM[1] = dM[3] = -((m2 + m3) * la * lb) * s1;
      else dM[5] = dM[7] = -(m3 * lb * lc) * s2;
      double T[9], R[9];
      mm_nn<3, 3, 3>(Minv, dM, T);
      mm_nn<3, 3, 3>(T, Minv, R);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fux[((3 + i) * NU + j) * NX + k] = -R[i * 3 + j];
    }
  }
  DEV static void f(const double *, const double *x, const double *u, double *xd) {
    const double la = 1.0, lb = 0.2, lc = 1.0, grav = 9.81;
    const double m1 = 1.0, m2 = 1.0, m3 = 0.5;
    double M[9];
    const double c1 = plant_cos(x[1]), c2 = plant_cos(x[2]), c12 = plant_cos(x[1] + x[2]);
    M[0] = (m1 + m2 + m3) * (la * la);
    M[4] = (m2 + m3) * (lb * lb);
    M[8] = m3 * (lc * lc);
    M[1] = M[3] = (m2 + m3) * la * lb * c1;
    M[5] = M[7] = m3 * lb * lc * c2;
    M[2] = M[6] = m3 * la * lc * c12;
    double G[3];
    G[0] = 0;
    G[1] = -(m2 + m3) * grav * lb * c1 - m3 * grav * lc * c12;
    G[2] = -m3 * grav * lc * c12;
    double Minv[9];
    inverse_pplu<3>(M, Minv);
    double rhs[3] = {u[0] - G[0], u[1] - G[1], u[2] - G[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      xd[i] = x[3 + i];
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += Minv[i * 3 + k] * rhs[k];
      xd[3 + i] = s;
    }
  }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    const double h = 2e-5;
    double xp[NX], up[NU], fp[NX], fm[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xp[i] = x[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) up[i] = u[i];
    for (int i = 0; i < NX; ++i) {
      xp[i] = x[i] + h; f(p, xp, u, fp);
      xp[i] = x[i] - h; f(p, xp, u, fm);
      for (int r = 0; r < NX; ++r) Fx[r * NX + i] = (fp[r] - fm[r]) / (2.0 * h);
      xp[i] = x[i];
    }
    for (int i = 0; i < NU; ++i) {
      up[i] = u[i] + h; f(p, x, up, fp);
      up[i] = u[i] - h; f(p, x, up, fm);
      for (int r = 0; r < NX; ++r) Fu[r * NU + i] = (fp[r] - fm[r]) / (2.0 * h);
      up[i] = u[i];
    }
  }
};

// ================================================================================ SYNTHETIC 7-joint manipulator, nx=14
struct Manip7Dyn {
  template <class S>
  DEV static void eval(const double *, const S *x, const S *u, S *xd) {
    const double mi[7] = {2.5, 2.0, 1.6, 1.2, 0.9, 0.6, 0.4};
    const double li[7] = {1.0, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3};
    const double wi[7] = {0.0, 1.4, 1.1, 0.8, 0.5, 0.3, 0.15};
    const double ci[7] = {0.0, 0.30, 0.25, 0.20, 0.15, 0.10, 0.05};
    const double grav = 9.81;
    // the 13 angles (7 cumulative, 6 neighbour differences) first, one batched cosine evaluation, then the joint rows
    S ang[13], sn[13], cs[13];
    S cum = S(0.0);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      cum = cum + x[i];
      ang[i] = cum;
      if (i > 0) ang[6 + i] = x[i] - x[i - 1];
    }
    trig_n<13, S>(ang, sn, cs);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      xd[i] = x[7 + i];
      S Mii = S(mi[i] * li[i] * li[i]);
      if (i > 0) { S cd = cs[6 + i]; Mii = Mii + S(ci[i]) * cd * cd; }
      S Gi = S(-grav * wi[i]) * cs[i];
      xd[7 + i] = (u[i] - Gi) / Mii;
    }
  }
};
struct Manip7Model {
  static constexpr int ID = CDDP_HIP_MODEL_MANIPULATOR7, NX = 14, NU = 7;
  static constexpr bool kDiscrete = false;
#ifdef CDDP_HOST_MODELS
  static constexpr bool kHasHess = true;
  static constexpr bool kHessBlocked = false;
  static void hess(const double *p, const double *x, const double *u, double *Fxx, double *Fuu, double *Fux) { ad_hessian<Manip7Dyn, NX, NU>(p, x, u, 1.0, Fxx, Fuu, Fux); }
#else
  static constexpr bool kHasHess = false;
  static constexpr bool kHessBlocked = true;   // see Quad12Model
  typedef Manip7Dyn HessDyn;
  static constexpr double kHessDiv = 1.0;
#endif
  DEV static void f(const double *p, const double *x, const double *u, double *xd) { Manip7Dyn::eval<double>(p, x, u, xd); }
  DEV static void jac(const double *p, const double *x, const double *u, double *Fx, double *Fu) {
    ad_jacobian<Manip7Dyn, NX, NU>(p, x, u, Fx, Fu);
  }
};

// ---- explicit integrators (dynamical_system.cpp:28-83) --------------------------------------
// Loop-invariant integrator constants held in registers (SGPRs) by the serial kernels: the step size products
// and the plant parameters would otherwise be re-fetched through dependent scalar loads -- and dt/6 re-divided --
// on every step of the chain.  Same values, same arithmetic as Stepper::step(integrator, dt, p, ...).
struct DynCtx {
  int integrator;
  double dt, hdt, dt6, dt2;
  double mp[32];
  DEV void load(int integrator_, double dt_, const double *mp_) {
    integrator = integrator_; dt = dt_; hdt = 0.5 * dt; dt6 = dt / 6; dt2 = 2 * dt;
#pragma unroll
    for (int i = 0; i < 32; ++i) mp[i] = mp_[i];   // unused entries are dead code after unrolling
  }
};

template <class Model, bool D = Model::kDiscrete> struct Stepper;
template <class Model> struct Stepper<Model, true> {
  DEV static void step(int, double, const double *p, const double *x, const double *u, double *xn) { Model::step(p, x, u, xn); }
  DEV static void step(const DynCtx &c, const double *x, const double *u, double *xn) { Model::step(c.mp, x, u, xn); }
};
template <class Model> struct Stepper<Model, false> {
  DEV static void step(int integrator, double dt, const double *p, const double *x, const double *u, double *xn) {
    constexpr int NX = Model::NX;
    double k1[NX];
    Model::f(p, x, u, k1);
    if (integrator == CDDP_HIP_EULER) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = x[i] + dt * k1[i];
      return;
    }
    double xt[NX], k2[NX];
    if (integrator == CDDP_HIP_HEUN) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k1[i];
      Model::f(p, xt, u, k2);
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = x[i] + (0.5 * dt) * (k1[i] + k2[i]);
      return;
    }
    double k3[NX];
    if (integrator == CDDP_HIP_RK3) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = x[i] + (0.5 * dt) * k1[i];
      Model::f(p, xt, u, k2);
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = (x[i] - dt * k1[i]) + (2 * dt) * k2[i];
      Model::f(p, xt, u, k3);
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = x[i] + (dt / 6) * ((k1[i] + 4.0 * k2[i]) + k3[i]);
      return;
    }
    double k4[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + (0.5 * dt) * k1[i];
    Model::f(p, xt, u, k2);
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + (0.5 * dt) * k2[i];
    Model::f(p, xt, u, k3);
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k3[i];
    Model::f(p, xt, u, k4);
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = x[i] + (dt / 6) * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
  }
  // the same integrators on hoisted constants
  DEV static void step(const DynCtx &c, const double *x, const double *u, double *xn) {
    constexpr int NX = Model::NX;
    const double *p = c.mp;
    double k1[NX];
    Model::f(p, x, u, k1);
    if (c.integrator == CDDP_HIP_EULER) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = x[i] + c.dt * k1[i];
      return;
    }
    double xt[NX], k2[NX];
    if (c.integrator == CDDP_HIP_HEUN) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = x[i] + c.dt * k1[i];
      Model::f(p, xt, u, k2);
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = x[i] + c.hdt * (k1[i] + k2[i]);
      return;
    }
    double k3[NX];
    if (c.integrator == CDDP_HIP_RK3) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = x[i] + c.hdt * k1[i];
      Model::f(p, xt, u, k2);
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = (x[i] - c.dt * k1[i]) + c.dt2 * k2[i];
      Model::f(p, xt, u, k3);
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = x[i] + c.dt6 * ((k1[i] + 4.0 * k2[i]) + k3[i]);
      return;
    }
    double k4[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + c.hdt * k1[i];
    Model::f(p, xt, u, k2);
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + c.hdt * k2[i];
    Model::f(p, xt, u, k3);
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + c.dt * k3[i];
    Model::f(p, xt, u, k4);
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = x[i] + c.dt6 * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
  }
};

}  // namespace cddp_dev
