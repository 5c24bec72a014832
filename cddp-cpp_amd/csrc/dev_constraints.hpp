// Device-side path constraints g(x,u) <= upper (reference include/cddp-cpp/cddp_core/constraint.hpp)
// and the quadratic objective (reference src/cddp_core/objective.cpp:30-154).
//
// The stacked dual layout of a problem (constraint objects in std::map order = sorted by name,
// ipddp_solver.cpp:1371-1384) is a COMPILE-TIME type list ConList<C1, C2, ...>, so that every
// index into the per-lane register arrays (y, s, g, Q_yx, Q_yu ...) is static after unrolling.
// Bounds / centres are run-time constants read through scalar loads from ProblemDev::pool.
#pragma once
#include "dev_linalg.hpp"
#include "dev_types.hpp"

namespace cddp_dev {

// BoxConstraint<Control> (constraint.hpp:144-251): g = [-u; u]*s - [-lb; ub]*s, G_u = [-I; I]*s
template <int D>
struct CtrlBox {
  static constexpr int KIND = CDDP_HIP_CON_CONTROL_BOX, DUAL = 2 * D, DIM = D;
  static constexpr bool HAS_X = false;   // G_x == 0
  static constexpr bool JAC_U = false;   // Jacobians do not depend on u
  template <int NX, int NU>
  DEV static void eval(const ConDev &c, const double *pool, const double *, const double *u, double *g) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      g[i] = (-u[i]) * c.scale - (-pool[c.off_lower + i]) * c.scale;
      g[D + i] = u[i] * c.scale - pool[c.off_upper + i] * c.scale;
    }
  }
  template <int NX, int NU>
  DEV static void jac(const ConDev &c, const double *, const double *, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int i = 0; i < D; ++i) { Gu[i * NU + i] = -c.scale; Gu[(D + i) * NU + i] = c.scale; }
    (void)Gx;
  }
  // loop-invariant constants in registers (serial kernels) + the same evaluation on them
  struct K { double scale, lo[D], hi[D]; };
  DEV static void load(const ConDev &c, const double *pool, K &k) {
    k.scale = c.scale;
#pragma unroll
    for (int i = 0; i < D; ++i) { k.lo[i] = pool[c.off_lower + i]; k.hi[i] = pool[c.off_upper + i]; }
  }
  template <int NX, int NU>
  DEV static void eval(const K &k, const double *, const double *u, double *g) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      g[i] = (-u[i]) * k.scale - (-k.lo[i]) * k.scale;
      g[D + i] = u[i] * k.scale - k.hi[i] * k.scale;
    }
  }
  template <int NX, int NU>
  DEV static void jac(const K &k, const double *, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int i = 0; i < D; ++i) { Gu[i * NU + i] = -k.scale; Gu[(D + i) * NU + i] = k.scale; }
    (void)Gx;
  }
};

// BoxConstraint<State>
template <int D>
struct StateBox {
  static constexpr int KIND = CDDP_HIP_CON_STATE_BOX, DUAL = 2 * D, DIM = D;
  static constexpr bool HAS_X = true;
  static constexpr bool JAC_U = false;
  template <int NX, int NU>
  DEV static void eval(const ConDev &c, const double *pool, const double *x, const double *, double *g) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      g[i] = (-x[i]) * c.scale - (-pool[c.off_lower + i]) * c.scale;
      g[D + i] = x[i] * c.scale - pool[c.off_upper + i] * c.scale;
    }
  }
  template <int NX, int NU>
  DEV static void jac(const ConDev &c, const double *, const double *, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int i = 0; i < D; ++i) { Gx[i * NX + i] = -c.scale; Gx[(D + i) * NX + i] = c.scale; }
    (void)Gu;
  }
  struct K { double scale, lo[D], hi[D]; };
  DEV static void load(const ConDev &c, const double *pool, K &k) {
    k.scale = c.scale;
#pragma unroll
    for (int i = 0; i < D; ++i) { k.lo[i] = pool[c.off_lower + i]; k.hi[i] = pool[c.off_upper + i]; }
  }
  template <int NX, int NU>
  DEV static void eval(const K &k, const double *x, const double *, double *g) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      g[i] = (-x[i]) * k.scale - (-k.lo[i]) * k.scale;
      g[D + i] = x[i] * k.scale - k.hi[i] * k.scale;
    }
  }
  template <int NX, int NU>
  DEV static void jac(const K &k, const double *, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int i = 0; i < D; ++i) { Gx[i * NX + i] = -k.scale; Gx[(D + i) * NX + i] = k.scale; }
    (void)Gu;
  }
};

// BallConstraint (constraint.hpp:313-404): g = -s*|x[:d]-c|^2 - (-(r*r)*s), G_x = -2 s (x-c)
template <int D>
struct Ball {
  static constexpr int KIND = CDDP_HIP_CON_BALL, DUAL = 1, DIM = D;
  static constexpr bool HAS_X = true;
  static constexpr bool JAC_U = false;
  template <int NX, int NU>
  DEV static void eval(const ConDev &c, const double *pool, const double *x, const double *, double *g) {
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) { double df = x[i] - pool[c.off_center + i]; sq += df * df; }
    g[0] = -(c.scale * sq) - (-(c.radius * c.radius) * c.scale);
  }
  template <int NX, int NU>
  DEV static void jac(const ConDev &c, const double *pool, const double *x, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int i = 0; i < D; ++i) Gx[i] = -2.0 * c.scale * (x[i] - pool[c.off_center + i]);
    (void)Gu;
  }
  struct K { double scale, radius, ctr[D]; };
  DEV static void load(const ConDev &c, const double *pool, K &k) {
    k.scale = c.scale; k.radius = c.radius;
#pragma unroll
    for (int i = 0; i < D; ++i) k.ctr[i] = pool[c.off_center + i];
  }
  template <int NX, int NU>
  DEV static void eval(const K &k, const double *x, const double *, double *g) {
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) { double df = x[i] - k.ctr[i]; sq += df * df; }
    g[0] = -(k.scale * sq) - (-(k.radius * k.radius) * k.scale);
  }
  template <int NX, int NU>
  DEV static void jac(const K &k, const double *x, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int i = 0; i < D; ++i) Gx[i] = -2.0 * k.scale * (x[i] - k.ctr[i]);
    (void)Gu;
  }
};

// LinearConstraint (constraint.hpp:253-311): g = A x - b
template <int R>
struct Linear {
  static constexpr int KIND = CDDP_HIP_CON_LINEAR, DUAL = R, DIM = R;
  static constexpr bool HAS_X = true;
  static constexpr bool JAC_U = false;
  template <int NX, int NU>
  DEV static void eval(const ConDev &c, const double *pool, const double *x, const double *, double *g) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += pool[c.off_A + r * NX + j] * x[j];
      g[r] = s - pool[c.off_b + r];
    }
  }
  template <int NX, int NU>
  DEV static void jac(const ConDev &c, const double *pool, const double *, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < NX; ++j) Gx[r * NX + j] = pool[c.off_A + r * NX + j];
    (void)Gu;
  }
  // (rows stay in the pool: R x NX doubles do not fit the scalar register file next to the plant constants)
  struct K { const double *A, *b; };
  DEV static void load(const ConDev &c, const double *pool, K &k) { k.A = pool + c.off_A; k.b = pool + c.off_b; }
  template <int NX, int NU>
  DEV static void eval(const K &k, const double *x, const double *, double *g) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += k.A[r * NX + j] * x[j];
      g[r] = s - k.b[r];
    }
  }
  template <int NX, int NU>
  DEV static void jac(const K &k, const double *, const double *, double *Gx, double *Gu) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < NX; ++j) Gx[r * NX + j] = k.A[r * NX + j];
    (void)Gu;
  }
};

// SecondOrderConeConstraint (constraint.hpp:626-800): g = cos(fov) sqrt(|p - o|^2 + eps) - (p - o) . axis, p = x[:3]; upper 0.
// Descriptor: center = cone origin (3), lower = unit opening direction (3; normalised by the caller as the reference's constructor
// does), radius = cos(fov), scale = eps.
struct SecondOrderCone {
  static constexpr int KIND = CDDP_HIP_CON_SOC, DUAL = 1, DIM = 3;
  static constexpr bool HAS_X = true;
  static constexpr bool JAC_U = false;
  struct K { double cosf, eps, o[3], ax[3]; };
  DEV static void load(const ConDev &c, const double *pool, K &k) {
    k.cosf = c.radius; k.eps = c.scale;
#pragma unroll
    for (int i = 0; i < 3; ++i) { k.o[i] = pool[c.off_center + i]; k.ax[i] = pool[c.off_lower + i]; }
  }
  template <int NX, int NU>
  DEV static void eval(const K &k, const double *x, const double *, double *g) {
    const double v0 = x[0] - k.o[0], v1 = x[1] - k.o[1], v2 = x[2] - k.o[2];
    const double reg_norm = sqrt(((v0 * v0 + v1 * v1) + v2 * v2) + k.eps);
    const double dot = (v0 * k.ax[0] + v1 * k.ax[1]) + v2 * k.ax[2];
    g[0] = reg_norm * k.cosf - dot;
  }
  template <int NX, int NU>
  DEV static void jac(const K &k, const double *x, const double *, double *Gx, double *Gu) {
    const double v[3] = {x[0] - k.o[0], x[1] - k.o[1], x[2] - k.o[2]};
    const double reg_norm = sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + k.eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) Gx[i] = (reg_norm > 1e-9) ? k.cosf * (v[i] / reg_norm) - k.ax[i] : -k.ax[i];
    (void)Gu;
  }
  template <int NX, int NU>
  DEV static void eval(const ConDev &c, const double *pool, const double *x, const double *u, double *g) { K k; load(c, pool, k); eval<NX, NU>(k, x, u, g); }
  template <int NX, int NU>
  DEV static void jac(const ConDev &c, const double *pool, const double *x, const double *u, double *Gx, double *Gu) { K k; load(c, pool, k); jac<NX, NU>(k, x, u, Gx, Gu); }
};

// ThrustMagnitudeConstraint (constraint.hpp:802-927; TWO = true: g = [min - |u|, |u| - max]) and MaxThrustMagnitudeConstraint
// (:929-1048; TWO = false: g = |u| - max); upper 0.  The value uses the plain norm, the Jacobian the regularised one
// u^T / sqrt(|u|^2 + eps), as the reference does.  Descriptor: radius = max, scale = eps, lower[0] = min (TWO only).
template <int D, bool TWO>
struct ThrustMagnitude {
  static constexpr int KIND = TWO ? CDDP_HIP_CON_THRUST : CDDP_HIP_CON_MAX_THRUST, DUAL = TWO ? 2 : 1, DIM = D;
  static constexpr bool HAS_X = false;
  static constexpr bool JAC_U = true;
  struct K { double mn, mx, eps; };
  DEV static void load(const ConDev &c, const double *pool, K &k) { k.mx = c.radius; k.eps = c.scale; k.mn = TWO ? pool[c.off_lower] : 0.0; }
  template <int NX, int NU>
  DEV static void eval(const K &k, const double *, const double *u, double *g) {
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) sq += u[i] * u[i];
    const double n = sqrt(sq);
    if (TWO) { g[0] = k.mn - n; g[1] = n - k.mx; } else g[0] = n - k.mx;
  }
  template <int NX, int NU>
  DEV static void jac(const K &k, const double *, const double *u, double *Gx, double *Gu) {
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) sq += u[i] * u[i];
    const double rn = sqrt(sq + k.eps);
    // :861-872 zeroes the rows when the regularised norm is below eps, :988-992 when it is not above DBL_MIN
    const bool ok = TWO ? !(rn < k.eps) : (rn > DBL_MIN);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const double d = ok ? u[i] / rn : 0.0;
      if (TWO) { Gu[i] = -d; Gu[NU + i] = d; } else Gu[i] = d;
    }
    (void)Gx;
  }
  template <int NX, int NU>
  DEV static void eval(const ConDev &c, const double *pool, const double *x, const double *u, double *g) { K k; load(c, pool, k); eval<NX, NU>(k, x, u, g); }
  template <int NX, int NU>
  DEV static void jac(const ConDev &c, const double *pool, const double *x, const double *u, double *Gx, double *Gu) { K k; load(c, pool, k); jac<NX, NU>(k, x, u, Gx, Gu); }
};

template <int OFF, int CI, class... Cs> struct ConImpl;
template <int OFF, int CI>
struct ConImpl<OFF, CI> {
  struct Ctx {};
  DEV static void load(const ProblemDev *, Ctx &) {}
  template <int NX, int NU> DEV static void eval(const Ctx &, const double *, const double *, double *) {}
  template <int NX, int NU> DEV static void jac(const Ctx &, const double *, const double *, double *, double *) {}
  template <int NX, int NU> DEV static void eval(const ProblemDev *, const double *, const double *, double *) {}
  template <int NX, int NU> DEV static void jac(const ProblemDev *, const double *, const double *, double *, double *) {}
};
template <int OFF, int CI, class C, class... Rest>
struct ConImpl<OFF, CI, C, Rest...> {
  typedef ConImpl<OFF + C::DUAL, CI + 1, Rest...> Next;
  struct Ctx { typename C::K k; typename Next::Ctx rest; };
  DEV static void load(const ProblemDev *P, Ctx &c) { C::load(P->cons[CI], P->pool, c.k); Next::load(P, c.rest); }
  template <int NX, int NU>
  DEV static void eval(const Ctx &c, const double *x, const double *u, double *g) {
    C::template eval<NX, NU>(c.k, x, u, g + OFF);
    Next::template eval<NX, NU>(c.rest, x, u, g);
  }
  template <int NX, int NU>
  DEV static void jac(const Ctx &c, const double *x, const double *u, double *Gx, double *Gu) {
    C::template jac<NX, NU>(c.k, x, u, Gx + OFF * NX, Gu + OFF * NU);
    Next::template jac<NX, NU>(c.rest, x, u, Gx, Gu);
  }
  template <int NX, int NU>
  DEV static void eval(const ProblemDev *P, const double *x, const double *u, double *g) {
    C::template eval<NX, NU>(P->cons[CI], P->pool, x, u, g + OFF);
    ConImpl<OFF + C::DUAL, CI + 1, Rest...>::template eval<NX, NU>(P, x, u, g);
  }
  template <int NX, int NU>
  DEV static void jac(const ProblemDev *P, const double *x, const double *u, double *Gx, double *Gu) {
    C::template jac<NX, NU>(P->cons[CI], P->pool, x, u, Gx + OFF * NX, Gu + OFF * NU);
    ConImpl<OFF + C::DUAL, CI + 1, Rest...>::template jac<NX, NU>(P, x, u, Gx, Gu);
  }
};

template <class... Cs>
struct ConList {
  static constexpr int NSEG = sizeof...(Cs);
  static constexpr int M = (0 + ... + Cs::DUAL);
  static constexpr bool HAS_X = (false || ... || Cs::HAS_X);   // any state-dependent constraint row
  static constexpr bool NEEDS_U = (false || ... || Cs::JAC_U); // some G_u depends on u (thrust-magnitude rows): call sites load u for jac()
  // segment table (constraint-major loops of computeTheta / computeBarrierMerit)
  DEV static int seg_dim(int c) { constexpr int dims[NSEG > 0 ? NSEG : 1] = {Cs::DUAL...}; return dims[c]; }
  DEV static int seg_off(int c) {
    constexpr int dims[NSEG > 0 ? NSEG : 1] = {Cs::DUAL...};
    int o = 0;
    for (int i = 0; i < c; ++i) o += dims[i];
    return o;
  }
  static bool matches(const ProblemDev &P) {   // host-side signature check
    constexpr int kinds[NSEG > 0 ? NSEG : 1] = {Cs::KIND...};
    constexpr int dims[NSEG > 0 ? NSEG : 1] = {Cs::DIM...};
    if (P.n_cons != NSEG) return false;
    for (int i = 0; i < NSEG; ++i) if (P.cons[i].kind != kinds[i] || P.cons[i].dim != dims[i]) return false;
    return true;
  }
  template <int NX, int NU>
  DEV static void eval(const ProblemDev *P, const double *x, const double *u, double *g) {
    ConImpl<0, 0, Cs...>::template eval<NX, NU>(P, x, u, g);
  }
  // hoisted-constant form for the serial kernels
  typedef typename ConImpl<0, 0, Cs...>::Ctx Ctx;
  DEV static void load(const ProblemDev *P, Ctx &c) { ConImpl<0, 0, Cs...>::load(P, c); }
  template <int NX, int NU>
  DEV static void eval(const Ctx &c, const double *x, const double *u, double *g) {
    ConImpl<0, 0, Cs...>::template eval<NX, NU>(c, x, u, g);
  }
  template <int NX, int NU>
  DEV static void jac(const Ctx &c, const double *x, const double *u, double *Gx, double *Gu) {
    ConImpl<0, 0, Cs...>::template jac<NX, NU>(c, x, u, Gx, Gu);
  }
  // Gx (M x NX) and Gu (M x NU) must be zero-filled by the caller
  template <int NX, int NU>
  DEV static void jac(const ProblemDev *P, const double *x, const double *u, double *Gx, double *Gu) {
    ConImpl<0, 0, Cs...>::template jac<NX, NU>(P, x, u, Gx, Gu);
  }
};
// Layouts whose only path constraint is a control box: every row of G_u has ONE entry (-s or +s) and G_x = 0.  The
// serial rollout uses this to form a row of G_u K as one product instead of a nu-term sum over known zeros.
template <class Cons> struct UDiag { static constexpr bool value = false; };
template <int D> struct UDiag<ConList<CtrlBox<D>>> {
  static constexpr bool value = true;
  static constexpr int col(int r) { return r < D ? r : r - D; }
  DEV static double val(const typename ConList<CtrlBox<D>>::Ctx &c, int r) { return r < D ? -c.k.scale : c.k.scale; }
};

template <>
struct ConList<> {
  static constexpr int NSEG = 0;
  static constexpr int M = 0;
  static constexpr bool HAS_X = false;
  static constexpr bool NEEDS_U = false;
  DEV static int seg_dim(int) { return 0; }
  DEV static int seg_off(int) { return 0; }
  static bool matches(const ProblemDev &P) { return P.n_cons == 0; }
  struct Ctx {};
  DEV static void load(const ProblemDev *, Ctx &) {}
  template <int NX, int NU> DEV static void eval(const Ctx &, const double *, const double *, double *) {}
  template <int NX, int NU> DEV static void jac(const Ctx &, const double *, const double *, double *, double *) {}
  template <int NX, int NU> DEV static void eval(const ProblemDev *, const double *, const double *, double *) {}
  template <int NX, int NU> DEV static void jac(const ProblemDev *, const double *, const double *, double *, double *) {}
};

// ---- QuadraticObjective (objective.cpp:80-154); Q_dt = Q*dt and R_dt = R*dt are stored in the pool
template <int NX, int NU>
struct Objective {
  DEV static void state_error(const ProblemDev *P, const double *xref_traj, int t, const double *x, double *e) {
    if (xref_traj) {   // per-step reference: constant-address-space scalar loads (see uniform_ptr)
      cptr_t r = uniform_ptr(xref_traj + (size_t)t * NX);
#pragma unroll
      for (int i = 0; i < NX; ++i) e[i] = x[i] - r[i];
    } else {           // fixed goal from the problem pool.  NOT through uniform_ptr: the integer round trip would
                       // capture the noalias ProblemDev argument and demote every later load from it to VMEM
#pragma unroll
      for (int i = 0; i < NX; ++i) e[i] = x[i] - P->pool[P->off_xref + i];
    }
  }
  // Loop-invariant cost matrices / goal in registers for the serial kernels (small plants only: NX*NX + NU*NU + NX
  // scalar-register doubles); same arithmetic as the pool-reading forms below.
  static constexpr bool kHoist = (NX * NX + NU * NU + NX) <= 40;
  struct Ctx { double Q[kHoist ? NX * NX : 1], R[kHoist ? NU * NU : 1], xr[kHoist ? NX : 1]; const double *Qp, *Rp, *xrp; };
  // Large plants (not hoisted): the serial kernels stage Q dt | R dt | x_ref in LDS (kStage doubles, stage()) and point the
  // context there.  Left on the problem pool the compiler hoists the loop-invariant scalar loads out of the step loop anyway,
  // runs out of SGPRs (nx = 12: 144 + 16 + 12 doubles = 344 SGPRs) and spills them to VGPR lanes: the rollout consumer of
  // the C4 quadrotor executed 861 v_readlane + 282 v_writelane per step, 30 % of its instruction stream.
  static constexpr int kStage = kHoist ? 1 : NX * NX + NU * NU + NX;
  DEV static void stage(const ProblemDev *P, double *lds, int tid, int nthreads) {
    if constexpr (!kHoist) {
      const double *Qp = P->pool + P->off_Qdt, *Rp = P->pool + P->off_Rdt, *xp = P->pool + P->off_xref;
      for (int e = tid; e < NX * NX; e += nthreads) lds[e] = Qp[e];
      for (int e = tid; e < NU * NU; e += nthreads) lds[NX * NX + e] = Rp[e];
      for (int e = tid; e < NX; e += nthreads) lds[NX * NX + NU * NU + e] = xp[e];
    }
  }
  DEV static void load_staged(const ProblemDev *P, Ctx &c, const double *lds) {
    if constexpr (kHoist) load(P, c);
    else { c.Qp = lds; c.Rp = lds + NX * NX; c.xrp = lds + NX * NX + NU * NU; }
  }
  DEV static void load(const ProblemDev *P, Ctx &c) {
    c.Qp = P->pool + P->off_Qdt; c.Rp = P->pool + P->off_Rdt; c.xrp = P->pool + P->off_xref;
    if constexpr (kHoist) {
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) c.Q[i] = c.Qp[i];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) c.R[i] = c.Rp[i];
#pragma unroll
      for (int i = 0; i < NX; ++i) c.xr[i] = c.xrp[i];
    }
  }
  DEV static double running_cost(const Ctx &c, const double *xref_traj, int t, const double *x, const double *u) {
    const double *Q = kHoist ? c.Q : c.Qp;
    const double *R = kHoist ? c.R : c.Rp;
    double e[NX];
    if (xref_traj) {
      cptr_t r = uniform_ptr(xref_traj + (size_t)t * NX);
#pragma unroll
      for (int i = 0; i < NX; ++i) e[i] = x[i] - r[i];
    } else {
      const double *xr = kHoist ? c.xr : c.xrp;
#pragma unroll
      for (int i = 0; i < NX; ++i) e[i] = x[i] - xr[i];
    }
    double sx = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double r = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) r += e[i] * Q[i * NX + j];
      sx += r * e[j];
    }
    double su = 0.0;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double r = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) r += u[i] * R[i * NU + j];
      su += r * u[j];
    }
    return sx + su;
  }
  // (e^T Q) e, row-vector-first association as `(e.transpose() * Q_ * e).value()`
  DEV static double running_cost(const ProblemDev *P, const double *xref_traj, int t, const double *x, const double *u) {
    double e[NX];
    state_error(P, xref_traj, t, x, e);
    const double *Q = P->pool + P->off_Qdt;
    const double *R = P->pool + P->off_Rdt;
    double sx = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double r = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) r += e[i] * Q[i * NX + j];
      sx += r * e[j];
    }
    double su = 0.0;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double r = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) r += u[i] * R[i * NU + j];
      su += r * u[j];
    }
    return sx + su;
  }
  DEV static double terminal_cost(const ProblemDev *P, const double *x) {
    double e[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) e[i] = x[i] - P->pool[P->off_xref + i];
    const double *Q = P->pool + P->off_Qf;
    double sx = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double r = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) r += e[i] * Q[i * NX + j];
      sx += r * e[j];
    }
    return sx;
  }
  // l_x = (2 Q_dt) e, l_u = (2 R_dt) u
  DEV static void lx(const ProblemDev *P, const double *xref_traj, int t, const double *x, double *out) {
    double e[NX];
    state_error(P, xref_traj, t, x, e);
    const double *Q = P->pool + P->off_Qdt;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += (2.0 * Q[i * NX + j]) * e[j];
      out[i] = s;
    }
  }
  DEV static void lu(const ProblemDev *P, const double *u, double *out) {
    const double *R = P->pool + P->off_Rdt;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NU; ++j) s += (2.0 * R[i * NU + j]) * u[j];
      out[i] = s;
    }
  }
  // V_x(N) = (2 Qf)(x_N - x_ref), V_xx(N) = 2 Qf
  DEV static void final_grad(const ProblemDev *P, const double *x, double *out) {
    const double *Q = P->pool + P->off_Qf;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) s += (2.0 * Q[i * NX + j]) * (x[j] - P->pool[P->off_xref + j]);
      out[i] = s;
    }
  }
};

}  // namespace cddp_dev
