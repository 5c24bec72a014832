// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the reference plants that the hot path calls through the
// DynamicalSystem plugin surface (reference src/cddp_core/dynamical_system.cpp and
// src/dynamics_model/*.cpp).  Each function cites the reference lines it follows.
//
// Jacobians follow each model's OWN derivative source in the reference:
//   Pendulum, Unicycle : analytic (pendulum.cpp:44-66, unicycle.cpp:44-66)
//   CartPole, Quadrotor: autodiff forward mode on getContinuousDynamicsAutodiff
//                        (cartpole.cpp:69-103, quadrotor.cpp:116-219) -- restated with the
//                        minimal forward-mode dual number below (autodiff v1.1.2 is a
//                        FetchContent dependency, CMakeLists.txt:116-125, absent here)
//   Manipulator        : central finite differences h=2e-5 (manipulator.cpp:53-70,
//                        helper.hpp:95-118)
//   LTISystem          : (A-I)/dt, B/dt (lti_system.cpp:78-92)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include "linalg.hpp"
#include "../cddp-cpp_amd/csrc/dev_trig.hpp"   // host build of the shared sin / cos (trig_mode); a header of the product read by the checker, never the reverse
#include "../include/cddp_hip.h"

namespace oracle {

// ---- minimal forward-mode dual number (first derivatives wrt up to 17 seeds) ------------
constexpr int kMaxSeeds = 24;
struct Dual {
  double v = 0.0;
  double d[kMaxSeeds];
  static int &np() { static thread_local int n = 0; return n; }
  Dual() { for (int i = 0; i < np(); ++i) d[i] = 0.0; }
  Dual(double x) : v(x) { for (int i = 0; i < np(); ++i) d[i] = 0.0; }
};
inline Dual operator+(const Dual &a, const Dual &b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < Dual::np(); ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
inline Dual operator-(const Dual &a, const Dual &b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < Dual::np(); ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
inline Dual operator-(const Dual &a) { Dual r; r.v = -a.v; for (int i = 0; i < Dual::np(); ++i) r.d[i] = -a.d[i]; return r; }
inline Dual operator*(const Dual &a, const Dual &b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < Dual::np(); ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
inline Dual operator/(const Dual &a, const Dual &b) {
  Dual r; r.v = a.v / b.v;
  for (int i = 0; i < Dual::np(); ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
  return r;
}
// libm-noise knob (test infrastructure, default off): with trig_noise() != 0 every sin / cos result is moved by -1 / 0 /
// +1 ulp, chosen by a hash of the argument.  glibc and the device libm both return sin / cos within an ulp but not the
// same bits; solves whose accept / reject decisions sit on last-bit knife edges (capped fraction-to-boundary trials,
// central-FD Jacobians with h = 2e-5) then follow different iterates.  The oracle-vs-noisy-oracle decision-flip rate
// (tests/test_oracle_trig_noise.py) is the yardstick the HIP-vs-oracle flip rates of tests/test_gpu_parity_r2.py are held to.
inline int &trig_noise() { static int v = 0; return v; }
inline double trig_perturb(double r, double a) {
  if (!trig_noise()) return r;
  std::uint64_t h; std::memcpy(&h, &a, sizeof(h));
  h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  switch (h % 3u) { case 0: return r; case 1: return std::nextafter(r, std::numeric_limits<double>::infinity());
                    default: return std::nextafter(r, -std::numeric_limits<double>::infinity()); }
}
// Shared-trig parity mode (test infrastructure, default off): with trig_mode() == 1 every sine / cosine goes through the
// branch-free routine the HIP library's parity build uses for the reference's plants (cddp-cpp_amd/csrc/dev_trig.hpp, compiled
// here for the host; 0.78 ulp against long-double libm, tests/test_dev_trig.py).  Both sides then execute the same IEEE
// add / mul / fma sequence per angle, so solves of the knife-edge plants are compared bit for bit instead of statistically
// (tests/test_shared_trig_parity.py).  The default (glibc) mode stays the independent check.
inline int &trig_mode() { static int v = 0; return v; }
inline int &failing_alpha_mask() { static int v = 0; return v; }   // test hook of Solver::performForwardPass (cddp_oracle.cpp), 0 = off
inline double base_sin(double a) { if (trig_mode() == 1) { double s, c; cddp_dev::sincos_1(a, &s, &c); return s; } return std::sin(a); }
inline double base_cos(double a) { if (trig_mode() == 1) { double s, c; cddp_dev::sincos_1(a, &s, &c); return c; } return std::cos(a); }
// log / pow of the solver core (barrier merit, barrier update, terminal-equality regularisation): glibc by default, the HIP parity
// build's straight-line routines in trig_mode 1
inline double olog(double x) { return trig_mode() == 1 ? cddp_dev::log_shared(x) : std::log(x); }
inline double opow(double x, double y) { return trig_mode() == 1 ? cddp_dev::pow_shared(x, y) : std::pow(x, y); }
inline double osin(double a) { return trig_perturb(base_sin(a), a); }
inline double ocos(double a) { return trig_perturb(base_cos(a), a + 0.5); }
inline double oasin(double a) { return trig_mode() == 1 ? cddp_dev::asin_shared(a) : std::asin(a); }
inline double otan(double a) { return trig_mode() == 1 ? base_sin(a) / base_cos(a) : std::tan(a); }   // parity build: sin / cos of the shared routine
inline double tan(double a) { return otan(a); }
inline double asin(double a) { return oasin(a); }
inline double sin(double a) { return osin(a); }   // found by the unqualified calls of the templated dynamics (S = double)
inline double cos(double a) { return ocos(a); }
inline Dual sin(const Dual &a) { Dual r; r.v = osin(a.v); double c = ocos(a.v); for (int i = 0; i < Dual::np(); ++i) r.d[i] = c * a.d[i]; return r; }
inline Dual cos(const Dual &a) { Dual r; r.v = ocos(a.v); double s = -osin(a.v); for (int i = 0; i < Dual::np(); ++i) r.d[i] = s * a.d[i]; return r; }
inline Dual sqrt(const Dual &a) { Dual r; r.v = std::sqrt(a.v); double g = 0.5 / r.v; for (int i = 0; i < Dual::np(); ++i) r.d[i] = g * a.d[i]; return r; }
inline Dual tan(const Dual &a) { Dual r; r.v = otan(a.v); double g = 1.0 + r.v * r.v; for (int i = 0; i < Dual::np(); ++i) r.d[i] = g * a.d[i]; return r; }
inline Dual asin(const Dual &a) { Dual r; r.v = oasin(a.v); double g = 1.0 / std::sqrt(1.0 - a.v * a.v); for (int i = 0; i < Dual::np(); ++i) r.d[i] = g * a.d[i]; return r; }

// Second-order forward mode (autodiff::dual2nd of the reference, dynamical_system.cpp:137-217, restated): value, gradient and
// Hessian w.r.t. up to kD2 seeded variables z = [x, u].  Only used for the plants whose Hessians the reference takes from
// autodiff (CartPole).
constexpr int kD2 = 21;   // quadrotor: 13 + 4 seeds; synthetic 7-joint arm: 14 + 7
struct Dual2 {
  double v = 0.0, d[kD2], h[kD2][kD2];
  static int &n() { static thread_local int k = 0; return k; }   // active seeds (set by Model::hessians)
  Dual2() { for (int i = 0; i < n(); ++i) { d[i] = 0.0; for (int j = 0; j < n(); ++j) h[i][j] = 0.0; } }
  Dual2(double x) : Dual2() { v = x; }
};
// r = phi(a) with phi', phi'' given:  grad = phi' a_d ; hess = phi'' a_d a_d^T + phi' a_h
inline Dual2 d2_unary(const Dual2 &a, double val, double p1, double p2) {
  Dual2 r; r.v = val;
  for (int i = 0; i < Dual2::n(); ++i) { r.d[i] = p1 * a.d[i]; for (int j = 0; j < Dual2::n(); ++j) r.h[i][j] = p2 * a.d[i] * a.d[j] + p1 * a.h[i][j]; }
  return r;
}
inline Dual2 operator+(const Dual2 &a, const Dual2 &b) { Dual2 r; r.v = a.v + b.v; for (int i = 0; i < Dual2::n(); ++i) { r.d[i] = a.d[i] + b.d[i]; for (int j = 0; j < Dual2::n(); ++j) r.h[i][j] = a.h[i][j] + b.h[i][j]; } return r; }
inline Dual2 operator-(const Dual2 &a, const Dual2 &b) { Dual2 r; r.v = a.v - b.v; for (int i = 0; i < Dual2::n(); ++i) { r.d[i] = a.d[i] - b.d[i]; for (int j = 0; j < Dual2::n(); ++j) r.h[i][j] = a.h[i][j] - b.h[i][j]; } return r; }
inline Dual2 operator-(const Dual2 &a) { Dual2 r; r.v = -a.v; for (int i = 0; i < Dual2::n(); ++i) { r.d[i] = -a.d[i]; for (int j = 0; j < Dual2::n(); ++j) r.h[i][j] = -a.h[i][j]; } return r; }
inline Dual2 operator*(const Dual2 &a, const Dual2 &b) {
  Dual2 r; r.v = a.v * b.v;
  for (int i = 0; i < Dual2::n(); ++i) {
    r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    for (int j = 0; j < Dual2::n(); ++j) r.h[i][j] = a.h[i][j] * b.v + a.d[i] * b.d[j] + a.d[j] * b.d[i] + a.v * b.h[i][j];
  }
  return r;
}
inline Dual2 d2_recip(const Dual2 &b) { const double inv = 1.0 / b.v; return d2_unary(b, inv, -inv * inv, 2.0 * inv * inv * inv); }
inline Dual2 operator/(const Dual2 &a, const Dual2 &b) { return a * d2_recip(b); }
inline Dual2 sin(const Dual2 &a) { const double s = osin(a.v), c = ocos(a.v); return d2_unary(a, s, c, -s); }
inline Dual2 cos(const Dual2 &a) { const double s = osin(a.v), c = ocos(a.v); return d2_unary(a, c, -s, -c); }
inline Dual2 sqrt(const Dual2 &a) { const double r = std::sqrt(a.v); return d2_unary(a, r, 0.5 / r, -0.25 / (a.v * r)); }
inline Dual2 asin(const Dual2 &a) { const double w = 1.0 - a.v * a.v, r = std::sqrt(w); return d2_unary(a, oasin(a.v), 1.0 / r, a.v / (w * r)); }
inline Dual2 tan(const Dual2 &a) { const double t = otan(a.v), g = 1.0 + t * t; return d2_unary(a, t, g, 2.0 * t * g); }

struct Model {
  int id = 0, nx = 0, nu = 0, integrator = 0;
  double dt = 0.0;
  double p[CDDP_HIP_MAX_MODEL_PARAMS];
  Mat A, B;  // LTI

  // ------------------------------------------------------------------ continuous dynamics
  template <typename S>
  void cartpole_f(const S *x, const S *u, S *xd, bool damping_term) const {
    // cartpole.cpp:38-67 (double path, no damping) and :69-103 (autodiff path, with damping)
    const double mc = p[0], mp = p[1], l = p[2], g = p[3], b = p[4];
    const S theta = x[1], x_dot = x[2], theta_dot = x[3], force = u[0];
    const S sin_theta = sin(theta), cos_theta = cos(theta);
    const double total_mass = mc + mp;
    const S den = S(mc) + S(mp) * sin_theta * sin_theta;
    xd[0] = x_dot;
    xd[1] = theta_dot;
    xd[2] = (force + S(mp) * sin_theta * (S(l) * theta_dot * theta_dot + S(g) * cos_theta)) / den;
    S num = -force * cos_theta - S(mp) * S(l) * theta_dot * theta_dot * cos_theta * sin_theta -
            S(total_mass) * S(g) * sin_theta;
    if (damping_term) num = num - S(b) * theta_dot;
    xd[3] = num / (S(l) * den);
  }

  template <typename S>
  void quadrotor_f(const S *x, const S *u, S *xd) const {
    // quadrotor.cpp:33-104 (double) == :166-219 (autodiff): same expression tree
    using std::sqrt;
    const double mass = p[0], arm = p[1], Ixx = p[2], Iyy = p[3], Izz = p[4], grav = p[5];
    for (int i = 0; i < 13; ++i) xd[i] = S(0.0);
    xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
    S qw = x[3], qx = x[4], qy = x[5], qz = x[6];
    S norm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    if (val(norm) > 1e-6) { qw = qw / norm; qx = qx / norm; qy = qy / norm; qz = qz / norm; }
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
    // R * [0,0,thrust]: third column of R (quadrotor.cpp:106-123)
    const S R02 = S(2.0) * (qx * qz + qy * qw);
    const S R12 = S(2.0) * (qy * qz - qx * qw);
    const S R22 = S(1.0) - S(2.0) * (qx * qx + qy * qy);
    const double invm = 1.0 / mass;
    xd[7] = S(invm) * (R02 * thrust);
    xd[8] = S(invm) * (R12 * thrust);
    xd[9] = S(invm) * (R22 * thrust) - S(grav);
    // inertia.inverse() for a fixed 3x3 = cofactors * (1/det) (Eigen compute_inverse_size3)
    const double c00 = Iyy * Izz, c11 = Ixx * Izz, c22 = Ixx * Iyy;
    const double det = c00 * Ixx;
    const double invdet = 1.0 / det;
    const double i00 = c00 * invdet, i11 = c11 * invdet, i22 = c22 * invdet;
    const S Iox = S(Ixx) * ox, Ioy = S(Iyy) * oy, Ioz = S(Izz) * oz;
    // omega x (I omega)
    const S cx = oy * Ioz - oz * Ioy;
    const S cy = oz * Iox - ox * Ioz;
    const S cz = ox * Ioy - oy * Iox;
    xd[10] = S(i00) * (tau_x - cx);
    xd[11] = S(i11) * (tau_y - cy);
    xd[12] = S(i22) * (tau_z - cz);
  }
  static double val(double v) { return v; }
  static double val(const Dual &v) { return v.v; }
  static double val(const Dual2 &v) { return v.v; }

  void manipulator_f(const double *x, const double *u, double *xd) const {
    // manipulator.cpp:29-51, 174-208 (la=1, lb=0.2, lc=1, g=9.81; manipulator.hpp:153-156)
    const double la = 1.0, lb = 0.2, lc = 1.0, grav = 9.81;
    const double m1 = 1.0, m2 = 1.0, m3 = 0.5;
    const double *q = x, *dq = x + 3;
    Mat M(3, 3);
    M(0, 0) = (m1 + m2 + m3) * (la * la);
    M(1, 1) = (m2 + m3) * (lb * lb);
    M(2, 2) = m3 * (lc * lc);
    M(0, 1) = M(1, 0) = (m2 + m3) * la * lb * ocos(q[1]);
    M(1, 2) = M(2, 1) = m3 * lb * lc * ocos(q[2]);
    M(0, 2) = M(2, 0) = m3 * la * lc * ocos(q[1] + q[2]);
    Vec G(3, 1);
    G(0) = 0;
    G(1) = -(m2 + m3) * grav * lb * ocos(q[1]) - m3 * grav * lc * ocos(q[1] + q[2]);
    G(2) = -m3 * grav * lc * ocos(q[1] + q[2]);
    Vec rhs(3, 1);
    for (int i = 0; i < 3; ++i) rhs(i) = u[i] - G(i);
    Vec ddq = inversePartialPivLU(M) * rhs;
    for (int i = 0; i < 3; ++i) { xd[i] = dq[i]; xd[3 + i] = ddq(i); }
  }

  // manipulator.cpp:210-275: the autodiff path (the base class's Hessian defaults differentiate THIS, dynamical_system.cpp:137-217):
  // ddq = M(q).inverse() * (tau - G(q)) with Eigen's generic inverse (PartialPivLU) in dual arithmetic
  template <typename S>
  void manipulator_f_ad(const S *x, const S *u, S *xd) const {
    const double la = 1.0, lb = 0.2, lc = 1.0, grav = 9.81, m1 = 1.0, m2 = 1.0, m3 = 0.5;
    const S cos_q1 = cos(x[1]), cos_q2 = cos(x[2]), cos_q12 = cos(x[1] + x[2]);
    S M[3][3];
    M[0][0] = S((m1 + m2 + m3) * (la * la)); M[1][1] = S((m2 + m3) * (lb * lb)); M[2][2] = S(m3 * (lc * lc));
    M[0][1] = M[1][0] = S((m2 + m3) * la * lb) * cos_q1;
    M[1][2] = M[2][1] = S(m3 * lb * lc) * cos_q2;
    M[0][2] = M[2][0] = S(m3 * la * lc) * cos_q12;
    S G[3];
    G[0] = S(0.0);
    G[1] = S(-(m2 + m3) * grav * lb) * cos(x[1]) - S(m3 * grav * lc) * cos_q12;
    G[2] = S(-m3 * grav * lc) * cos_q12;
    // PartialPivLU of M (row pivoting on |value|), then inverse = U^-1 L^-1 P applied to the right-hand side
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
      int piv = k; double best = std::fabs(val2(M[k][k]));
      for (int r = k + 1; r < 3; ++r) if (std::fabs(val2(M[r][k])) > best) { best = std::fabs(val2(M[r][k])); piv = r; }
      if (piv != k) { for (int c = 0; c < 3; ++c) std::swap(M[k][c], M[piv][c]); std::swap(perm[k], perm[piv]); }
      for (int r = k + 1; r < 3; ++r) {
        M[r][k] = M[r][k] / M[k][k];
        for (int c = k + 1; c < 3; ++c) M[r][c] = M[r][c] - M[r][k] * M[k][c];
      }
    }
    S rhs[3], y[3], z[3];
    for (int i = 0; i < 3; ++i) rhs[i] = u[i] - G[i];
    for (int i = 0; i < 3; ++i) { y[i] = rhs[perm[i]]; for (int c = 0; c < i; ++c) y[i] = y[i] - M[i][c] * y[c]; }
    for (int i = 2; i >= 0; --i) { z[i] = y[i]; for (int c = i + 1; c < 3; ++c) z[i] = z[i] - M[i][c] * z[c]; z[i] = z[i] / M[i][i]; }
    for (int i = 0; i < 3; ++i) { xd[i] = x[3 + i]; xd[3 + i] = z[i]; }
  }
  static double val2(double v) { return v; }
  static double val2(const Dual2 &v) { return v.v; }
  static double val2(const Dual &v) { return v.v; }

  // bicycle.cpp:28-45 (double) == :47-66 (autodiff); params: wheelbase; state [x, y, theta, v], control [a, delta]
  template <typename S>
  void bicycle_f(const S *x, const S *u, S *xd) const {
    const double L = p[0];
    const S theta = x[2], v = x[3], a = u[0], delta = u[1];
    xd[0] = v * cos(theta);
    xd[1] = v * sin(theta);
    xd[2] = (v / S(L)) * tan(delta);
    xd[3] = a;
  }

  // car.cpp:24-60 (getDiscreteDynamics, double) and :164-216 (getDiscreteDynamicsAutodiff: the same tree plus the two clamps);
  // params: wheelbase; state [x, y, theta, v], control [steering delta, acceleration a]; a DISCRETE plant: h = timestep
  template <typename S>
  void car_next(const S *x, const S *u, S *xn, bool clamps) const {
    using std::sqrt;
    const double d = p[0], h = dt;
    const S theta = x[2], v = x[3], delta = u[0], a = u[1];
    const S cos_theta = cos(theta), sin_theta = sin(theta);
    const S f = S(h) * v;
    const S f_sin_delta = f * sin(delta);
    S inside = S(d * d) - f_sin_delta * f_sin_delta;
    if (clamps && val2(inside) < 0.0) inside = S(0.0);
    const S b = S(d) + f * cos(delta) - sqrt(inside);
    S asin_arg = sin(delta) * f / S(d);
    if (clamps && std::fabs(val2(asin_arg)) > 1.0) asin_arg = S(val2(asin_arg) > 0.0 ? 1.0 : -1.0);
    const S dtheta = asin(asin_arg);
    xn[0] = x[0] + b * cos_theta;
    xn[1] = x[1] + b * sin_theta;
    xn[2] = x[2] + dtheta;
    xn[3] = x[3] + S(h) * a;
  }

  // SYNTHETIC (not in the reference): BASELINE config 4 shape nx=12 -- the same rigid-body
  // quadrotor with ZYX Euler attitude, state [p(3), v(3), phi,theta,psi, omega(3)].
  template <typename S>
  void quad12_f(const S *x, const S *u, S *xd) const {
    const double mass = p[0], arm = p[1], Ixx = p[2], Iyy = p[3], Izz = p[4], grav = p[5];
    const S phi = x[6], th = x[7], psi = x[8];
    const S ox = x[9], oy = x[10], oz = x[11];
    const S sph = sin(phi), cph = cos(phi), sth = sin(th), cth = cos(th), sps = sin(psi), cps = cos(psi);
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

  // SYNTHETIC (not in the reference): BASELINE config 5 shape nx=14/nu=7 -- 7-joint
  // generalisation of the simplified manipulator with a diagonal mass matrix so that
  // M^{-1} is closed form: ddq_i = (tau_i - G_i(q)) / M_ii,
  // M_ii = m_i * l_i^2 (+ coupling weight c_i * cos(q_i - q_{i-1})^2 kept >= 0),
  // G_i = -g * w_i * cos(sum_{j<=i} q_j).  Constants are fixed here and mirrored in the kernel.
  template <typename S>
  void manip7_f(const S *x, const S *u, S *xd) const {
    static const double mi[7] = {2.5, 2.0, 1.6, 1.2, 0.9, 0.6, 0.4};
    static const double li[7] = {1.0, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3};
    static const double wi[7] = {0.0, 1.4, 1.1, 0.8, 0.5, 0.3, 0.15};
    static const double ci[7] = {0.0, 0.30, 0.25, 0.20, 0.15, 0.10, 0.05};
    const double grav = 9.81;
    S cum = S(0.0);
    for (int i = 0; i < 7; ++i) {
      xd[i] = x[7 + i];
      cum = cum + x[i];
      S Mii = S(mi[i] * li[i] * li[i]);
      if (i > 0) { S cd = cos(x[i] - x[i - 1]); Mii = Mii + S(ci[i]) * cd * cd; }
      S Gi = S(-grav * wi[i]) * cos(cum);
      xd[7 + i] = (u[i] - Gi) / Mii;
    }
  }

  void f(const double *x, const double *u, double /*time*/, double *xd) const {
    switch (id) {
      case CDDP_HIP_MODEL_PENDULUM: {
        // pendulum.cpp:29-42 (+sin convention; theta=0 upright)
        const double length = p[0], mass = p[1], damping = p[2], gravity = p[3];
        const double inertia = mass * length * length;
        xd[0] = x[1];
        xd[1] = (u[0] - damping * x[1] + mass * gravity * length * osin(x[0])) / inertia;
        break;
      }
      case CDDP_HIP_MODEL_CARTPOLE: cartpole_f<double>(x, u, xd, false); break;
      case CDDP_HIP_MODEL_UNICYCLE: {
        // unicycle.cpp:28-42
        xd[0] = u[0] * ocos(x[2]);
        xd[1] = u[0] * osin(x[2]);
        xd[2] = u[1];
        break;
      }
      case CDDP_HIP_MODEL_LTI: {
        // DynamicalSystem::getContinuousDynamics default (dynamical_system.cpp:85-99):
        // (A x + B u - x) / dt.  Only used if an integrator is forced on an LTI system.
        for (int i = 0; i < nx; ++i) {
          double s = 0; for (int j = 0; j < nx; ++j) s += A(i, j) * x[j];
          double t = 0; for (int j = 0; j < nu; ++j) t += B(i, j) * u[j];
          xd[i] = ((s + t) - x[i]) / dt;
        }
        break;
      }
      case CDDP_HIP_MODEL_QUADROTOR: quadrotor_f<double>(x, u, xd); break;
      case CDDP_HIP_MODEL_MANIPULATOR: manipulator_f(x, u, xd); break;
      case CDDP_HIP_MODEL_BICYCLE: bicycle_f<double>(x, u, xd); break;
      case CDDP_HIP_MODEL_HCW: {   // spacecraft_linear.cpp:32-54; params: mean_motion, mass
        const double n = p[0], n2 = n * n, mass = p[1];
        xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
        xd[3] = 2.0 * n * x[4] + 3.0 * n2 * x[0] + u[0] / mass;
        xd[4] = -2.0 * n * x[3] + u[1] / mass;
        xd[5] = -n2 * x[2] + u[2] / mass;
        break;
      }
      case CDDP_HIP_MODEL_CAR: {   // DynamicalSystem::getContinuousDynamics default (dynamical_system.cpp:85-99): (x+ - x) / dt
        double xn[4]; car_next<double>(x, u, xn, false);
        for (int i = 0; i < 4; ++i) xd[i] = (xn[i] - x[i]) / dt;
        break;
      }
      case CDDP_HIP_MODEL_QUADROTOR_EULER12: quad12_f<double>(x, u, xd); break;
      case CDDP_HIP_MODEL_MANIPULATOR7: manip7_f<double>(x, u, xd); break;
      default: std::fprintf(stderr, "oracle: unknown model %d\n", id); std::abort();
    }
  }

  // ------------------------------------------------ discrete dynamics (dynamical_system.cpp:28-83)
  Vec step(const Vec &x, const Vec &u, double time) const {
    if (id == CDDP_HIP_MODEL_LTI) return A * x + B * u;  // lti_system.cpp:71-76
    if (id == CDDP_HIP_MODEL_CAR) { Vec xn(4, 1); car_next<double>(x.a, u.a, xn.a, false); return xn; }   // car.cpp:24-60 overrides getDiscreteDynamics
    const int n = nx;
    auto F = [&](const Vec &xx, double tt) { Vec k(n, 1); f(xx.a, u.a, tt, k.a); return k; };
    switch (integrator) {
      case CDDP_HIP_EULER: return x + dt * F(x, time);
      case CDDP_HIP_HEUN: {
        Vec k1 = F(x, time);
        Vec k2 = F(x + dt * k1, time + dt);
        return x + (0.5 * dt) * (k1 + k2);
      }
      case CDDP_HIP_RK3: {
        Vec k1 = F(x, time);
        Vec k2 = F(x + (0.5 * dt) * k1, time + 0.5 * dt);
        Vec k3 = F(x - dt * k1 + (2 * dt) * k2, time + dt);
        return x + (dt / 6) * (k1 + 4.0 * k2 + k3);
      }
      case CDDP_HIP_RK4: {
        Vec k1 = F(x, time);
        Vec k2 = F(x + (0.5 * dt) * k1, time + 0.5 * dt);
        Vec k3 = F(x + (0.5 * dt) * k2, time + 0.5 * dt);
        Vec k4 = F(x + dt * k3, time + dt);
        return x + (dt / 6) * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
      }
    }
    return Vec::Zero(n);  // "Integration type not supported!" -> zeros (:79-82)
  }

  // ------------------------------------------------ continuous-time Jacobians f_x, f_u
  template <typename FN>
  void ad_jac(FN fn, const Vec &x, const Vec &u, Mat &Fx, Mat &Fu) const {
    Dual::np() = nx + nu;
    std::vector<Dual> xs(nx), us(nu), xd(nx);
    for (int i = 0; i < nx; ++i) { xs[i] = Dual(x(i)); xs[i].d[i] = 1.0; }
    for (int j = 0; j < nu; ++j) { us[j] = Dual(u(j)); us[j].d[nx + j] = 1.0; }
    fn(xs.data(), us.data(), xd.data());
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) Fx(i, j) = xd[i].d[j];
      for (int j = 0; j < nu; ++j) Fu(i, j) = xd[i].d[nx + j];
    }
    Dual::np() = 0;
  }

  // ------------------------------------------------ continuous-time Hessian tensors f_xx[i] (nx x nx), f_uu[i] (nu x nu),
  // f_ux[i] (nu x nx), each with the reference's own source: analytic overrides where the model has them, the autodiff
  // default (dynamical_system.cpp:137-217) on getContinuousDynamicsAutodiff otherwise.  Returns false for plants without
  // a restated Hessian (use_ilqr = false is then refused).
  template <typename FN>
  void ad_hess(FN fn, const Vec &x, const Vec &u, std::vector<Mat> &Fxx, std::vector<Mat> &Fuu, std::vector<Mat> &Fux, double div = 1.0) const {
    Dual2::n() = nx + nu;
    std::vector<Dual2> xs(nx), us(nu), xd(nx);
    for (int i = 0; i < nx; ++i) { xs[i] = Dual2(x(i)); xs[i].d[i] = 1.0; }
    for (int j = 0; j < nu; ++j) { us[j] = Dual2(u(j)); us[j].d[nx + j] = 1.0; }
    fn(xs.data(), us.data(), xd.data());
    for (int i = 0; i < nx; ++i) {
      for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) Fxx[i](a, b) = xd[i].h[a][b] / div;
      for (int a = 0; a < nu; ++a) for (int b = 0; b < nu; ++b) Fuu[i](a, b) = xd[i].h[nx + a][nx + b] / div;
      for (int a = 0; a < nu; ++a) for (int b = 0; b < nx; ++b) Fux[i](a, b) = xd[i].h[nx + a][b] / div;
    }
    Dual2::n() = 0;
  }

  bool hessians(const Vec &x, const Vec &u, double /*time*/, std::vector<Mat> &Fxx, std::vector<Mat> &Fuu, std::vector<Mat> &Fux) const {
    Fxx.assign(nx, Mat::Zero(nx, nx)); Fuu.assign(nx, Mat::Zero(nu, nu)); Fux.assign(nx, Mat::Zero(nu, nx));
    switch (id) {
      case CDDP_HIP_MODEL_PENDULUM: {   // pendulum.cpp:68-85 (state: analytic, control: zero); the cross Hessian is autodiff of the
        const double length = p[0], gravity = p[3];   // -sin twin (:87-100), whose u-x second derivatives vanish
        Fxx[1](0, 0) = -(gravity / length) * osin(x(0));
        return true;
      }
      case CDDP_HIP_MODEL_UNICYCLE: {   // unicycle.cpp:68-89 analytic state / zero control Hessian; cross Hessian = autodiff of :91-107
        Fxx[0](2, 2) = -u(0) * ocos(x(2));
        Fxx[1](2, 2) = -u(0) * osin(x(2));
        Fux[0](0, 2) = -osin(x(2));
        Fux[1](0, 2) = ocos(x(2));
        return true;
      }
      case CDDP_HIP_MODEL_LTI: return true;   // lti_system.cpp:94-115: zero
      case CDDP_HIP_MODEL_HCW: return true;   // spacecraft_linear.cpp:85-120: zero
      case CDDP_HIP_MODEL_CARTPOLE:           // cartpole.cpp:191-199 -> DynamicalSystem defaults (dual2nd through the autodiff path)
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { cartpole_f<Dual2>(xs, us, xd, true); }, x, u, Fxx, Fuu, Fux);
        return true;
      case CDDP_HIP_MODEL_BICYCLE: {          // bicycle.cpp:113-156 analytic state / control Hessians; cross = base default (autodiff)
        std::vector<Mat> sx, su;
        sx.assign(nx, Mat::Zero(nx, nx)); su.assign(nx, Mat::Zero(nu, nu));
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { bicycle_f<Dual2>(xs, us, xd); }, x, u, sx, su, Fux);
        const double L = p[0], theta = x(2), v = x(3), delta = u(1);
        Fxx[0](2, 2) = -v * ocos(theta); Fxx[0](2, 3) = -osin(theta); Fxx[0](3, 2) = -osin(theta);
        Fxx[1](2, 2) = -v * osin(theta); Fxx[1](2, 3) = ocos(theta); Fxx[1](3, 2) = ocos(theta);
        const double c = ocos(delta);
        Fuu[2](1, 1) = 2.0 * v * osin(delta) / (L * (c * c * c));   // std::pow(cos, 3) in the reference
        return true;
      }
      case CDDP_HIP_MODEL_CAR:                // car.cpp:113-161: hessian of the discrete map / timestep; cross = base default on (x+ - x) / dt
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { car_next<Dual2>(xs, us, xd, true); }, x, u, Fxx, Fuu, Fux, dt);
        return true;
      case CDDP_HIP_MODEL_MANIPULATOR: {      // manipulator.cpp:72-86: state / control Hessians are ZERO overrides; cross = base default
        std::vector<Mat> sx, su;
        sx.assign(nx, Mat::Zero(nx, nx)); su.assign(nx, Mat::Zero(nu, nu));
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { manipulator_f_ad<Dual2>(xs, us, xd); }, x, u, sx, su, Fux);
        return true;
      }
      case CDDP_HIP_MODEL_QUADROTOR:          // quadrotor.cpp:224-278: dual2nd through the normalised-quaternion dynamics.  The reference's
        // getCrossHessian returns nx x nu matrices (a Jacobian over u of the x-gradient) where the solvers add nu x nx blocks
        // (ipddp_solver.cpp:1070-1082): an Eigen size mismatch, i.e. use_ilqr = false is not defined behaviour for this plant there.
        // Restated with the block the solver's formula needs (d2 f_i / du dx, nu x nx).
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { quadrotor_f<Dual2>(xs, us, xd); }, x, u, Fxx, Fuu, Fux);
        return true;
      case CDDP_HIP_MODEL_QUADROTOR_EULER12:  // synthetic plants: no overrides, i.e. the base class's dual2nd default on the plant's own expression
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { quad12_f<Dual2>(xs, us, xd); }, x, u, Fxx, Fuu, Fux);
        return true;
      case CDDP_HIP_MODEL_MANIPULATOR7:
        ad_hess([&](const Dual2 *xs, const Dual2 *us, Dual2 *xd) { manip7_f<Dual2>(xs, us, xd); }, x, u, Fxx, Fuu, Fux);
        return true;
      default: return false;
    }
  }

  void jacobians(const Vec &x, const Vec &u, double time, Mat &Fx, Mat &Fu) const {
    Fx = Mat(nx, nx); Fu = Mat(nx, nu);
    switch (id) {
      case CDDP_HIP_MODEL_PENDULUM: {
        // pendulum.cpp:44-66
        const double length = p[0], mass = p[1], damping = p[2], gravity = p[3];
        Fx(0, 1) = 1.0;
        Fx(1, 0) = (gravity / length) * ocos(x(0));
        Fx(1, 1) = -damping / (mass * length * length);
        Fu(1, 0) = 1.0 / (mass * length * length);
        break;
      }
      case CDDP_HIP_MODEL_CARTPOLE:
        ad_jac([&](const Dual *xs, const Dual *us, Dual *xd) { cartpole_f<Dual>(xs, us, xd, true); }, x, u, Fx, Fu);
        break;
      case CDDP_HIP_MODEL_UNICYCLE: {
        // unicycle.cpp:44-66
        Fx(0, 2) = -u(0) * osin(x(2));
        Fx(1, 2) = u(0) * ocos(x(2));
        Fu(0, 0) = ocos(x(2));
        Fu(1, 0) = osin(x(2));
        Fu(2, 1) = 1.0;
        break;
      }
      case CDDP_HIP_MODEL_LTI: {
        // lti_system.cpp:78-92
        Mat Am = A; for (int i = 0; i < nx; ++i) Am(i, i) -= 1.0;
        Fx = Am / dt; Fu = B / dt;
        break;
      }
      case CDDP_HIP_MODEL_QUADROTOR:
        ad_jac([&](const Dual *xs, const Dual *us, Dual *xd) { quadrotor_f<Dual>(xs, us, xd); }, x, u, Fx, Fu);
        break;
      case CDDP_HIP_MODEL_HCW: {       // spacecraft_linear.cpp:56-83
        const double n = p[0], n2 = n * n, mass = p[1];
        Fx(0, 3) = 1.0; Fx(1, 4) = 1.0; Fx(2, 5) = 1.0;
        Fx(3, 0) = 3.0 * n2; Fx(3, 4) = 2.0 * n; Fx(4, 3) = -2.0 * n; Fx(5, 2) = -n2;
        Fu(3, 0) = 1.0 / mass; Fu(4, 1) = 1.0 / mass; Fu(5, 2) = 1.0 / mass;
        break;
      }
      case CDDP_HIP_MODEL_BICYCLE: {   // bicycle.cpp:68-111 analytic
        const double L = p[0], theta = x(2), v = x(3), delta = u(1);
        Fx(0, 2) = -v * osin(theta); Fx(0, 3) = ocos(theta);
        Fx(1, 2) = v * ocos(theta); Fx(1, 3) = osin(theta);
        Fx(2, 3) = otan(delta) / L;
        Fu(3, 0) = 1.0;
        const double c = ocos(delta);
        Fu(2, 1) = v / (L * (c * c));   // std::pow(cos, 2)
        break;
      }
      case CDDP_HIP_MODEL_CAR: {       // car.cpp:62-111: autodiff of the DISCRETE map, J.diagonal() -= 1, J /= timestep
        ad_jac([&](const Dual *xs, const Dual *us, Dual *xd) { car_next<Dual>(xs, us, xd, true); }, x, u, Fx, Fu);
        for (int i = 0; i < nx; ++i) Fx(i, i) -= 1.0;
        Fx = Fx / dt; Fu = Fu / dt;
        break;
      }
      case CDDP_HIP_MODEL_QUADROTOR_EULER12:
        ad_jac([&](const Dual *xs, const Dual *us, Dual *xd) { quad12_f<Dual>(xs, us, xd); }, x, u, Fx, Fu);
        break;
      case CDDP_HIP_MODEL_MANIPULATOR7:
        ad_jac([&](const Dual *xs, const Dual *us, Dual *xd) { manip7_f<Dual>(xs, us, xd); }, x, u, Fx, Fu);
        break;
      case CDDP_HIP_MODEL_MANIPULATOR: {
        // finite_difference_jacobian, central, h = 2e-5 (helper.hpp:95-118)
        const double h = 2e-5;
        Vec xp = x;
        Vec fp(nx, 1), fm(nx, 1);
        for (int i = 0; i < nx; ++i) {
          xp(i) = x(i) + h; f(xp.a, u.a, time, fp.a);
          xp(i) = x(i) - h; f(xp.a, u.a, time, fm.a);
          for (int r = 0; r < nx; ++r) Fx(r, i) = (fp(r) - fm(r)) / (2.0 * h);
          xp(i) = x(i);
        }
        Vec up = u;
        for (int i = 0; i < nu; ++i) {
          up(i) = u(i) + h; f(x.a, up.a, time, fp.a);
          up(i) = u(i) - h; f(x.a, up.a, time, fm.a);
          for (int r = 0; r < nx; ++r) Fu(r, i) = (fp(r) - fm(r)) / (2.0 * h);
          up(i) = u(i);
        }
        break;
      }
      default: std::fprintf(stderr, "oracle: unknown model %d\n", id); std::abort();
    }
  }
};

}  // namespace oracle
