// Lane-cooperative form of the stack-fed sweep (stacks.hip) for plug-ins with a large state: sixteen lanes per trajectory,
// every matrix of a step in LDS, each output element owned by ONE lane that accumulates it in exactly the order the
// one-lane kernel `sweep<NX, NU, M>` uses (k ascending from 0.0) -- the two kernels are bitwise interchangeable
// (tests/test_stack_fed_coop.py).  The one-lane form keeps ~6 nx^2 doubles per lane live: at nx = 12 it runs from scratch
// memory (10-18 KB per lane) and takes 155-200 ms per sweep of 2048 trajectories (profiles/r02_stackfed_sweeps.md); here a
// trajectory's step state is 12-17 KB of LDS shared by its sixteen lanes and nothing spills.
//
// Same branches as the one-lane kernel (clddp_solver.cpp:79-204, ipddp_solver.cpp:1048-1118 / 1355-1568, logddp_solver.cpp:470-575,
// msipddp_solver.cpp:1112-1208).  The small replicated pieces (nu x nu factorisation, BoxQP, the scalar reductions) are done by every
// lane of the group on identical inputs, so "failed" is group-uniform without a vote.
// Included by stacks.hip inside its anonymous namespace (uses StackArgs, SI, clipp, clips).
#pragma once

DEV void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// blocks of one 64-trajectory tile on one XCD: the sixteen single-wave workgroups of a tile read 32-B pieces of the same
// 128-B lines of the batch-minor stacks (same reasoning as kernels_coop.hpp::coop_group, profiles/r03_pmc_calibration_32B.md)
DEV int sc_group(int bid) {
  const int sup = bid >> 7, r = bid & 127;
  return (sup * 8 + (r & 7)) * 16 + (r >> 3);
}
inline unsigned sc_grid(int B) { return (unsigned)(((B + 3) / 4 + 127) / 128 * 128); }

template <int NX, int NU, int M>
struct SCfg {
  static constexpr int MM = M > 0 ? M : 1;
  enum : int {
    oVxx = 0, oVx = oVxx + NX * NX, oA = oVx + NX, oB = oA + NX * NX, oW = oB + NX * NU, oQx = oW + NX, oQu = oQx + NX,
    oQxx = oQu + NU, oQux = oQxx + NX * NX, oQuu = oQux + NU * NX, oT1 = oQuu + NU * NU, oT2 = oT1 + NX * NX, oFree = oT2 + NU * NX,
    // (round 6) Overlays, by the order a step touches its areas -- T1 = A^T V and T2 = B^T V are dead once the Q blocks exist, f_u once Q_uu
    // does, R_x before V_n (which also overlays T1) is written, and nothing reads the reduction row inside the sweep:
    //   gains K in T2, K^T Q_uu in f_u's area, regularised Q_uu / R_u / R_x in T1 (where they fit), the reduction row in Q_xx's area.
    // 389 doubles less per trajectory at nx = 14, nu = 7, m = 14: three workgroups per CU instead of two.
    oKK = oT2, oKtQ = oB,
    kOvl = (NU * NX + NU * NU + NU <= NX * NX) ? 1 : 0,
    oQr = kOvl ? oT1 : oFree, oRu = oQr + NU * NU, oRx = oRu + NU,
    oY = kOvl ? oFree : oRx + NU * NX, oS = oY + MM, oGg = oS + MM, oGx = oGg + MM, oGu = oGx + MM * NX, oYS = oGu + MM * NU, oSir = oYS + MM,
    oRhat = oSir + MM, oRp = oRhat + MM, oSs = oRp + MM, oEnd = oSs + MM,
    kRedOvl = (NX * NX >= 32) ? 1 : 0, oRed = kRedOvl ? oQxx : oEnd, SIZE0 = kRedOvl ? oEnd : oEnd + 32,
    STRIDE = SIZE0 | 1   // odd: the four trajectories of a wavefront start in different LDS banks
  };
  static constexpr int oVn = oT1;   // V_xx before symmetrisation overlays T1 (dead after the A-products)
};

// "for e in my elements of [0, E)": element e of a step quantity belongs to lane e mod 16 of the trajectory's group
#define SC_EACH(E, e) _Pragma("unroll") for (int e##_it = 0; e##_it < ((E) + 15) / 16; ++e##_it) if (const int e = e##_it * 16 + gl; e < (E))
// the same for the matrix-sized quantities, rolled: unrolled, the scheduler hoists every LDS read of the phase and the kernel
// needs all 512 registers plus scratch
// "for c in my columns of an NC-column matrix"
// Global element e of step t of an [t][E][Bp] stack: "global_load v, v_off, s[base]" with ONE wave-uniform 64-bit base per (stack, step)
// (SGPR pair: stack + t E Bp) and a 32-bit lane offset that does not depend on the stack or the step -- element e = e_it * 16 + gl of
// trajectory b sits at ((e_it * 16 + gl) Bp + b) * 8 behind the base: the offsets vo.v[e_it] are formed once per kernel and kept opaque
// (round 6: with the element's uniform part folded into the base instead, every access carried its own s_add / s_addc / shift chain --
// ~1 900 of the 7 000 instructions of a step at nx = 12; a wavefront alone on its SIMD issues one instruction per ~5 cycles, so the
// instruction count is the step time).  Written as base + ((t * E + e) * Bp + b) the compiler strength-reduces EVERY access into its
// own 64-bit VGPR pointer that lives across the whole sweep (~70 pairs: the first version spilled 250 registers).
// (address space 1: the laundered pointer would otherwise be a FLAT access, which counts in lgkmcnt as well -- every lds_sync() behind
// the step's prefetch then waited for the prefetch itself)
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) char gchar;
// the kernel's argument block, as the kernel-argument segment holds it (k_stacks_backward_coop has ONE parameter, at offset 0)
typedef const __attribute__((address_space(4))) StackArgs *sc_kargs_t;
template <int KMAX>
struct SCOff {
  // Two layouts (StackArgs::t4).  Plain [t][e][Bp]: element e of trajectory b sits (e Bp + b) 8 bytes behind the step's base.  Tile-minor
  // [t][b / 4][e][b % 4]: (e 4 + b % 4) 8 bytes behind base + (b / 4) E 32 -- the four trajectories of the workgroup share a contiguous
  // E x 32-byte record, and a load of sixteen consecutive elements is 512 contiguous bytes.  In both, a step is E Bp doubles.
  unsigned v[KMAX], b8;   // byte offset of the lane's element gl / of the trajectory inside an element
  int t4, goff;           // layout; (b / 4) * 4 under t4 (the record's offset is E * goff doubles), else 0
  DEV void init(int gl, int tl, int b, int Bp, int t4_) {
    t4 = t4_; goff = t4_ ? (b >> 2) * 4 : 0;
    b8 = t4_ ? (unsigned)tl * 8u : (unsigned)b * 8u;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      v[k] = t4_ ? ((unsigned)(k * 16 + gl) * 4u + (unsigned)tl) * 8u : ((unsigned)(k * 16 + gl) * (unsigned)Bp + (unsigned)b) * 8u;
      asm volatile("" : "+v"(v[k]));
    }
  }
};
template <int NX, int NU, int M> struct SCOffFor { typedef SCOff<1> type; };
// One buffer resource per (stack, step): base = stack + t E Bp in an SGPR quad, no stride, no range to speak of; every access is then ONE
// instruction, buffer_load / buffer_store dwordx2 v, v_off, s[rsrc], s_off offen.  (global_load with a 64-bit SGPR base + 32-bit VGPR
// offset would do as well, but the zero-extension of the loop-invariant offsets is hoisted out of the step loop and the instruction
// selector then sees a 64-bit VGPR add per access.)  Offsets are 32-bit: the host side keeps the cooperative form to E Bp 8 < 2^32.
typedef unsigned int sc_u32x2 __attribute__((ext_vector_type(2)));
DEV __amdgpu_buffer_rsrc_t sc_rsrc(const double *base, int t, int E, int Bp, int goff) {
  const unsigned long long v = (unsigned long long)(base + (size_t)t * E * Bp + (size_t)E * goff);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffull)), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), (short)0, (int)0xffffffff, 0x00020000);
}
DEV double sc_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
DEV void sc_st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sc_u32x2, x), r, voff, soff, 0); }
// element e of an SC_EACH(E, e) loop
// (the lane offset of element 0 in the VGPR offset, the sixteen-element stride e_it * 16 Bp 8 in the instruction's SCALAR offset: one offset
//  register for the whole kernel -- a table of per-e_it offsets was 9 - 13 more, and at nx >= 13 they were spilled and came back through
//  scratch loads whose vmcnt(0) waits serialised the prefetch stream)
#define SC_ES8 (vo.t4 ? 32u : (unsigned)bpo * 8u)   /* bytes from one element to the next */
#define SC_RS(stack, t, E) sc_rsrc((stack), (t), (E), bpo, vo.goff)
#define SC_LD(stack, t, E, e) sc_ld(SC_RS(stack, t, E), vo.v[0], (unsigned)(e##_it * 16) * SC_ES8)
#define SC_ST(stack, t, E, e, x) sc_st(SC_RS(stack, t, E), vo.v[0], (unsigned)(e##_it * 16) * SC_ES8, (x))
// element eu + gl, eu uniform (a column owner walking rows): the uniform part rides in the scalar offset
#define SC_LDU(stack, t, E, eu) sc_ld(SC_RS(stack, t, E), vo.v[0], (unsigned)(eu) * SC_ES8)
#define SC_STU(stack, t, E, eu, x) sc_st(SC_RS(stack, t, E), vo.v[0], (unsigned)(eu) * SC_ES8, (x))
// an element index that is NOT "uniform + gl" (a row owner walking its row, the replicated scalar pieces): per-lane 32-bit offset
#define SC_LDV(stack, t, E, e) sc_ld(SC_RS(stack, t, E), (unsigned)(e) * SC_ES8 + vo.b8, 0u)
#define SC_STV(stack, t, E, e, x) sc_st(SC_RS(stack, t, E), (unsigned)(e) * SC_ES8 + vo.b8, 0u, (x))
// The step index and the batch pitch are laundered through an empty asm once per step: otherwise the optimiser turns every access into
// its own loop-carried 64-bit induction variable (~70 SGPR pairs, spilled to VGPR lanes: 628 v_readlane / v_writelane in the step
// loop); laundered, the address arithmetic stays inside the step on the otherwise idle scalar unit.
#define SC_OPAQUE(tt, bpo, t) int tt = (t), bpo = a.Bp; asm volatile("" : "+s"(tt), "+s"(bpo))
#define SC_COLS(NC, c) for (int c = gl; c < (NC); c += 16)
#define SC_LOOP(E, e) _Pragma("unroll 3") for (int e = gl; e < (E); e += 16)

template <int NU>
struct SCFactor {   // the factorisation the one-lane kernel uses for this size, on register copies
  LDLTs<NU> f;
  DEV bool compute(const double *Q) { f.compute(Q, NU); return f.ok; }
  DEV void solve(double *x) const { f.solve(x); }
};
template <>
struct SCFactor<2> {
  LDLTd<2> f;
  DEV bool compute(const double *Q) { f.compute(Q, 2); return f.ok; }
  DEV void solve(double *x) const { f.solve(x); }
};
template <>
struct SCFactor<1> {
  double d;
  DEV bool compute(const double *Q) { d = Q[0]; return true; }
  DEV void solve(double *x) const { x[0] = ldlt1_solve(d, x[0]); }
};

// One step's input record, spread over the sixteen lanes of a group (element e on lane e mod 16), in registers between its
// fetch (issued a step ahead) and the moment the LDS areas of the previous step are free.
template <int NX, int NU, int M>
struct SCRec {
  typedef SCfg<NX, NU, M> C;
  static constexpr int cdiv(int x) { return (x + 15) / 16; }
  static constexpr int MM = M > 0 ? M : 1;
  double A[cdiv(NX * NX)], Bm[cdiv(NX * NU)], lx[cdiv(NX)], lu[cdiv(NU)], lxx[cdiv(NX * NX)], luu[cdiv(NU * NU)], lux[cdiv(NU * NX)];
  double y[cdiv(MM)], s[cdiv(MM)], g[cdiv(MM)], Gx[cdiv(MM * NX)], Gu[cdiv(MM * NU)];
  // (three groups: fetched / parked group by group in the middle of a step -- each behind the last reader of its LDS areas -- was tried to
  //  shorten the record's register life and measured slower at nx = 12, 28.4 k -> 33.8 k clocks per step: the VMEM issue of a group is
  //  only hidden behind a long stretch of arithmetic; and it does not combine with K^T Q_uu living in f_u's area)
  template <class AT, class VO> DEV void fetchA(const AT &a, int t0, int gl, const VO &vo) {
    SC_OPAQUE(t, bpo, t0);
    SC_EACH(NX * NX, e) A[e_it] = SC_LD(a.fx, t, NX * NX, e);
    SC_EACH(NX * NU, e) Bm[e_it] = SC_LD(a.fu, t, NX * NU, e);
  }
  template <class AT, class VO> DEV void fetchC(const AT &a, int t0, int gl, const VO &vo) {
    if constexpr (M > 0) {
      SC_OPAQUE(t, bpo, t0);
      SC_EACH(M, e) { y[e_it] = SC_LD(a.y, t, M, e); s[e_it] = SC_LD(a.s, t, M, e); g[e_it] = SC_LD(a.g, t, M, e); }
      SC_EACH(M * NX, e) Gx[e_it] = SC_LD(a.Gx, t, M * NX, e);
      SC_EACH(M * NU, e) Gu[e_it] = SC_LD(a.Gu, t, M * NU, e);
    }
  }
  template <class AT, class VO> DEV void fetchQ(const AT &a, int t0, int gl, const VO &vo) {
    SC_OPAQUE(t, bpo, t0);
    SC_EACH(NX, e) lx[e_it] = SC_LD(a.lx, t, NX, e);
    SC_EACH(NU, e) lu[e_it] = SC_LD(a.lu, t, NU, e);
    SC_EACH(NX * NX, e) lxx[e_it] = SC_LD(a.lxx, t, NX * NX, e);
    SC_EACH(NU * NU, e) luu[e_it] = SC_LD(a.luu, t, NU * NU, e);
    SC_EACH(NU * NX, e) lux[e_it] = SC_LD(a.lux, t, NU * NX, e);
  }
  template <class AT, class VO> DEV void fetch(const AT &a, int t0, int gl, const VO &vo) { fetchA(a, t0, gl, vo); fetchC(a, t0, gl, vo); fetchQ(a, t0, gl, vo); }
  DEV void park(double *__restrict__ L, int gl) const { parkA(L, gl); parkC(L, gl); parkQ(L, gl); }
  DEV void parkA(double *__restrict__ L, int gl) const {
    SC_EACH(NX * NX, e) L[C::oA + e] = A[e_it];
    SC_EACH(NX * NU, e) L[C::oB + e] = Bm[e_it];
  }
  DEV void parkC(double *__restrict__ L, int gl) const {
    if constexpr (M > 0) {
      SC_EACH(M, e) { L[C::oY + e] = y[e_it]; L[C::oS + e] = s[e_it]; L[C::oGg + e] = g[e_it]; }
      SC_EACH(M * NX, e) L[C::oGx + e] = Gx[e_it];
      SC_EACH(M * NU, e) L[C::oGu + e] = Gu[e_it];
    }
  }
  DEV void parkQ(double *__restrict__ L, int gl) const {
    SC_EACH(NX, e) L[C::oQx + e] = lx[e_it];
    SC_EACH(NU, e) L[C::oQu + e] = lu[e_it];
    SC_EACH(NX * NX, e) L[C::oQxx + e] = lxx[e_it];
    SC_EACH(NU * NU, e) L[C::oQuu + e] = luu[e_it];
    SC_EACH(NU * NX, e) L[C::oQux + e] = lux[e_it];
  }
};

// R rows of  out(i) = sum_k L[addr(i, k)] * v[k]  (k ascending from 0.0): the operand rows are double-buffered by hand -- the LDS
// reads of rows i + 2, i + 3 are in flight while rows i, i + 1 are reduced (two independent chains).  With one wavefront per SIMD
// nothing else hides the LDS round trip: the counters of the first version showed the wave waiting 60 % of its cycles
// (profiles/r03_stackfed_sweeps.md).  sched_barrier keeps the compiler from hoisting every read of the phase to its top (which
// needs all 512 registers plus scratch).
template <int R, int K, class Addr, class Term, class Out>
DEV void sc_rows_t(const double *__restrict__ L, Addr addr, Term term, Out out) {
  double buf[2][2][K];
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (h < R) {
#pragma unroll
      for (int k = 0; k < K; ++k) buf[0][h][k] = L[addr(h, k)];
    }
#pragma unroll
  for (int i = 0; i < R; i += 2) {
    const int cur = (i >> 1) & 1, nxt = cur ^ 1;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (i + 2 + h < R) {
#pragma unroll
        for (int k = 0; k < K; ++k) buf[nxt][h][k] = L[addr(i + 2 + h, k)];
      }
    __builtin_amdgcn_sched_barrier(0);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      s0 += term(k, buf[cur][0][k]);
      if (i + 1 < R) s1 += term(k, buf[cur][1][k]);
    }
    out(i, s0);
    if (i + 1 < R) out(i + 1, s1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int R, int K, class Addr, class Out>
DEV void sc_rows(const double *__restrict__ L, const double (&v)[K], Addr addr, Out out) {
  sc_rows_t<R, K>(L, addr, [&](int k, double x) { return x * v[k]; }, out);
}

// ---------------------------------------------------------------------------------------------------------------- register-tiled products
// Round 6.  A wavefront that is alone on its SIMD pays every LDS round trip in full, and the compiler cannot move an LDS read above an
// LDS write through the same pointer: a section written "for each output: read operands, reduce, write" is one exposed round trip per
// output (the step loop carried ~490 s_waitcnt; profiles/r06_stackfed_coop.md).  Every section below is therefore "all reads -> registers,
// arithmetic, all writes", and the matrix products are tiled: the R x CC outputs are cut into BR x BC blocks, one per lane of the group,
// so that a lane reads (BR + BC) operands per k for BR BC multiply-adds (one column per lane needs 1 + 1 / rows reads per multiply-add
// and leaves 16 - CC lanes idle).  Each output element is still ONE sequential sum over k ascending from 0.0 of the same products.
#ifndef SC_V1_NX
#define SC_V1_NX 13   // state sizes from here on keep the column-per-lane step (sweep_coop_v1)
#endif
#ifndef SC_FETCH_LATE
#define SC_FETCH_LATE 0
#endif
#ifndef SC_MM_BUDGET
#define SC_MM_BUDGET 24   // doubles of operands per chunk of a tiled product (two chunks in flight)
#endif
constexpr int sc_tile_pick(int R, int CC, bool tall) {   // BR * 256 + BC: fewest outputs per lane with at most sixteen blocks, then the fewest operand reads
  int best = 256 * R + CC, bestArea = 1 << 30, bestSum = 1 << 30;
  for (int br = 1; br <= R; ++br)
    for (int bc = 1; bc <= CC; ++bc) {
      if (((R + br - 1) / br) * ((CC + bc - 1) / bc) > 16) continue;
      const int area = br * bc, sum = br + bc;
      const bool better = area < bestArea || (area == bestArea && (sum < bestSum || (sum == bestSum && tall && br > bc)));
      if (better) { best = br * 256 + bc; bestArea = area; bestSum = sum; }
    }
  return best;
}
template <int R, int CC, bool TALL = false>
struct SCTile {
  static constexpr int pk = sc_tile_pick(R, CC, TALL), BR = pk >> 8, BC = pk & 255, NBR = (R + BR - 1) / BR, NBC = (CC + BC - 1) / BC;
  static constexpr bool kAll = NBR * NBC == 16;
  int ri[BR], cj[BC];   // LDS operand indices (clamped into the matrix: a lane past the edge reads a valid element and writes nothing)
  bool vr[BR], vc[BC];
  DEV void init(int gl) {
    const int bi0 = gl / NBC, bj = gl - bi0 * NBC;
    const bool act = kAll || bi0 < NBR;
    const int bi = kAll ? bi0 : (bi0 < NBR ? bi0 : NBR - 1);
#pragma unroll
    for (int r = 0; r < BR; ++r) { const int i = bi * BR + r; vr[r] = act && ((R % BR) == 0 || i < R); ri[r] = ((R % BR) == 0 || i < R) ? i : R - 1; }
#pragma unroll
    for (int q = 0; q < BC; ++q) { const int c = bj * BC + q; vc[q] = (CC % BC) == 0 || c < CC; cj[q] = ((CC % BC) == 0 || c < CC) ? c : CC - 1; }
  }
};
struct SCNoMid { DEV int operator()(int) const { return 0; } };
// out(i, c, sum_{k < K} lhs(i, k) [* mid(k)] * rhs(k, c), pre(i, c)) for the lane's block; lhs / rhs / mid give LDS indices, pre(i, c) a value that
// is read BEFORE anything is written (the old value of an updated entry, ...), out does the writes
template <int R, int CC, int K, bool MID, bool TALL = false, class LA, class RA, class MA, class Pre, class Out>
DEV void sc_mm(const double *__restrict__ L, const int gl, LA lhs, RA rhs, MA mid, Pre pre, Out out) {
  typedef SCTile<R, CC, TALL> T;
  constexpr int BR = T::BR, BC = T::BC;
  constexpr int KC0 = SC_MM_BUDGET / (BR + BC + (MID ? 1 : 0)), KC = KC0 < 1 ? 1 : (KC0 > K ? K : KC0), NCH = (K + KC - 1) / KC;
  T tl; tl.init(gl);
  decltype(pre(0, 0)) pv[BR][BC];
  double acc[BR][BC];
#pragma unroll
  for (int r = 0; r < BR; ++r)
#pragma unroll
    for (int q = 0; q < BC; ++q) { pv[r][q] = pre(tl.ri[r], tl.cj[q]); acc[r][q] = 0.0; }
  struct Chunk { double a[KC][BR], b[KC][BC], m[KC]; };
  auto ldc = [&](const int ch, Chunk &c) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int k = ch * KC + kk;
      if (k < K) {
#pragma unroll
        for (int r = 0; r < BR; ++r) c.a[kk][r] = L[lhs(tl.ri[r], k)];
#pragma unroll
        for (int q = 0; q < BC; ++q) c.b[kk][q] = L[rhs(k, tl.cj[q])];
        if (MID) c.m[kk] = L[mid(k)];
      }
    }
  };
  auto cmp = [&](const int ch, const Chunk &c) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int k = ch * KC + kk;
      if (k < K) {
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const double am = MID ? c.a[kk][r] * c.m[kk] : c.a[kk][r];
#pragma unroll
          for (int q = 0; q < BC; ++q) acc[r][q] += am * c.b[kk][q];
        }
      }
    }
  };
  Chunk c0, c1;
  ldc(0, c0);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) { if ((ch & 1) == 0) ldc(ch + 1, c1); else ldc(ch + 1, c0); }
    __builtin_amdgcn_sched_barrier(0);
    if ((ch & 1) == 0) cmp(ch, c0); else cmp(ch, c1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < BR; ++r)
#pragma unroll
    for (int q = 0; q < BC; ++q)
      if (tl.vr[r] && tl.vc[q]) out(tl.ri[r], tl.cj[q], acc[r][q], pv[r][q]);
}

#ifdef SC_TIMING   // experiment: cycles per section of the step loop, summed over the sweep, per workgroup (profiles/scripts/sc_times.py)
__device__ unsigned long long g_sc_times[4096 * 16];
#define SC_TICK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); tk_acc[k] += now_ - tk_last; tk_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SC_TICK(k) do { } while (0)
#endif
// =====================================================================================================================================
// The step as rounds 3 - 5 wrote it (one column of a product per lane, operands double-buffered two rows at a time), kept for nx >= 13: the
// tiled step below needs more registers than a wavefront has there (nx = 14, nu = 7, m = 14: ~210 scratch accesses per step, 13 -> 27 ms
// per sweep; profiles/r06_stackfed_coop.md), and at nx = 13 the two measure alike.  Same arithmetic, same addressing macros.
template <int NX, int NU, int M>
DEV bool sweep_coop_v1(const StackArgs &a0, const int b, const int gl, const typename SCOffFor<NX, NU, M>::type &vo, double *__restrict__ L, const double reg,
                    const double mu, double &dV0, double &dV1, double &inf_du, double &inf_pr, double &inf_comp, double &step_norm) {
  typedef SCfg<NX, NU, M> C;
  const StackArgs &a = a0;
  const int N = a.N, bpo = a.Bp;
  const bool lg = a.branch == CDDP_HIP_STACKS_LOGDDP;
  const bool msp = a.branch == CDDP_HIP_STACKS_MSIPDDP_PATH;
  const bool ms = a.branch == CDDP_HIP_STACKS_MSIPDDP || msp;
  const bool ip = a.branch != CDDP_HIP_STACKS_CLDDP && !lg;
  SC_EACH(NX, i) L[C::oVx + i] = a.VxN[(size_t)i * a.Bp + b];
  if (ip || lg) {
    SC_LOOP(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      L[C::oVxx + e] = 0.5 * (a.VxxN[(size_t)(i * NX + c) * a.Bp + b] + a.VxxN[(size_t)(c * NX + i) * a.Bp + b]);
    }
  } else {
    SC_LOOP(NX * NX, e) L[C::oVxx + e] = a.VxxN[(size_t)e * a.Bp + b];
  }
  lds_sync();
  SC_EACH(NX, i) SC_ST(a.Vx, N, NX, i, L[C::oVx + i]);
  SC_EACH(NX * NX, e) SC_ST(a.Vxx, N, NX * NX, e, L[C::oVxx + e]);
  dV0 = dV1 = 0.0; inf_du = inf_pr = inf_comp = step_norm = 0.0;
  double norm_Vx = 0.0;
  if (!ip && !lg) {
#pragma unroll
    for (int i = 0; i < NX; ++i) norm_Vx += fabs(L[C::oVx + i]);
  }
  SCRec<NX, NU, M> rec;
  rec.fetch(a, N - 1, gl, vo);
#ifdef SC_TIMING
  unsigned long long tk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_last = __builtin_readcyclecounter();
#endif
  const bool have_Fxx = a0.Fxx != nullptr;
  for (int t_ = N - 1; t_ >= 0; --t_) {
    SC_OPAQUE(t, bpo, t_);
    // The ~45 stack pointers of the argument block do not fit the scalar registers next to the step's buffer resources: kept across
    // the loop they were parked in VGPR lanes and came back through ~390 v_readlane per step.  Read through a pointer the optimiser
    // cannot see through, they are re-fetched from the kernel-argument segment (scalar cache) where a step needs them.
    sc_kargs_t kp = (sc_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const auto &a = *kp;
    SC_TICK(15);
    // ---------------------------------------------------------------- step record -> LDS (fetched during the previous step)
    rec.park(L, gl);
    // w = V_x, or V_x + V_xx d_t under multiple shooting (V of step t + 1 is already in LDS)
    if (ms) {
      SC_EACH(NX, i) {
        double s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s1 += L[C::oVxx + i * NX + k] * SC_LDV(a.dfc, t, NX, k);
        L[C::oW + i] = L[C::oVx + i] + s1;
      }
    } else {
      SC_EACH(NX, i) L[C::oW + i] = L[C::oVx + i];
    }
    lds_sync();
    SC_TICK(0);
    if (t > 0) rec.fetch(a, t - 1, gl, vo);   // in flight behind this step's arithmetic
    // ---------------------------------------------------------------- Q_x, Q_u, T1 = A^T V_xx, T2 = B^T V_xx
    SC_EACH(NX, i) {
      double q = L[C::oQx + i];
      if constexpr (M > 0) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGx + r * NX + i] * L[C::oY + r];
        q = q + s1;
      }
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s2 += L[C::oA + k * NX + i] * L[C::oW + k];
      L[C::oQx + i] = q + s2;
    }
    SC_EACH(NU, i) {
      double q = L[C::oQu + i];
      if constexpr (M > 0) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGu + r * NU + i] * L[C::oY + r];
        q = q + s1;
      }
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s2 += L[C::oB + k * NU + i] * L[C::oW + k];
      L[C::oQu + i] = q + s2;
    }
    SC_TICK(1);
    // column owners: lane c keeps column c of the right-hand operand in registers and walks the rows; the other operand arrives
    // as group-wide broadcast reads (one LDS address per trajectory and instruction)
    SC_COLS(NX, c) {
      double v[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) v[k] = L[C::oVxx + k * NX + c];
      sc_rows<NX, NX>(L, v, [](int i, int k) { return C::oA + k * NX + i; }, [&](int i, double s1) { L[C::oT1 + i * NX + c] = s1; });
      sc_rows<NU, NX>(L, v, [](int i, int k) { return C::oB + k * NU + i; }, [&](int i, double s1) { L[C::oT2 + i * NX + c] = s1; });
    }
    lds_sync();
    SC_TICK(2);
    // ---------------------------------------------------------------- Q_xx += T1 A, Q_ux += T2 A, Q_uu += T2 B (+ tensor terms)
    SC_COLS(NX, c) {
      double v[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) v[k] = L[C::oA + k * NX + c];
      sc_rows<NX, NX>(L, v, [](int i, int k) { return C::oT1 + i * NX + k; }, [&](int i, double s1) {
        const int e = i * NX + c;
        double q = L[C::oQxx + e] + s1;
        if (have_Fxx) for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * SC_LDU(a.Fxx, t, NX * NX * NX, j * NX * NX + i * NX);
        L[C::oQxx + e] = q;
      });
      sc_rows<NU, NX>(L, v, [](int i, int k) { return C::oT2 + i * NX + k; }, [&](int i, double s1) {
        const int e = i * NX + c;
        double q = L[C::oQux + e] + s1;
        if (have_Fxx) for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * SC_LDU(a.Fux, t, NX * NU * NX, j * NU * NX + i * NX);
        L[C::oQux + e] = q;
      });
    }
    SC_EACH(NU * NU, e) {
      const int i = e / NU, c = e - i * NU;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s1 += L[C::oT2 + i * NX + k] * L[C::oB + k * NU + c];
      double q = L[C::oQuu + e] + s1;
      if (have_Fxx) for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * SC_LDU(a.Fuu, t, NX * NU * NU, j * NU * NU + e_it * 16);
      L[C::oQuu + e] = q;
    }
    lds_sync();
    SC_TICK(3);
    // ---------------------------------------------------------------- gains
    double kk[NU];
    if constexpr (M > 0) {
      const double s_floor = dmax(mu * 1e-3, kEpsSlackS);
      SC_EACH(M, r) {
        const double y = L[C::oY + r], s = L[C::oS + r], g = L[C::oGg + r];
        const double ssafe = msp ? s : dmax(s, s_floor);
        const double rp = g + s;
        const double rc = y * s - mu;
        const double rhat = y * rp - rc;
        L[C::oSs + r] = ssafe; L[C::oYS + r] = msp ? y / s : clipp(y, ssafe); L[C::oRp + r] = rp; L[C::oRhat + r] = rhat;
        L[C::oSir + r] = msp ? rhat / s : clips(rhat, ssafe);
        inf_pr = dmax(inf_pr, fabs(rp)); inf_comp = dmax(inf_comp, fabs(rc));
      }
      lds_sync();
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += (L[C::oGu + r * NU + i] * L[C::oYS + r]) * L[C::oGu + r * NU + c];
        double q = 0.5 * (L[C::oQuu + i * NU + c] + L[C::oQuu + c * NU + i]) + s1;
        if (i == c) q += reg;
        L[C::oQr + e] = q;
      }
      SC_EACH(NU, i) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGu + r * NU + i] * L[C::oSir + r];
        L[C::oRu + i] = L[C::oQu + i] + s1;
      }
      SC_COLS(NX, c) {
        double gxc[M], ys[M];
#pragma unroll
        for (int r = 0; r < M; ++r) { gxc[r] = L[C::oGx + r * NX + c]; ys[r] = L[C::oYS + r]; }
        sc_rows_t<NU, M>(L, [](int i, int r) { return C::oGu + r * NU + i; }, [&](int r, double x) { return (x * ys[r]) * gxc[r]; },
                         [&](int i, double s2) { L[C::oRx + i * NX + c] = L[C::oQux + i * NX + c] + s2; });
      }
      lds_sync();
      SC_TICK(4);
      {
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQr + i];
        SCFactor<NU> f;
        if (!f.compute(Qr)) return false;
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = L[C::oRu + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = L[C::oRx + i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = -col[i];
        }
      }
      lds_sync();
      SC_TICK(5);
      // slack / dual direction gains (:1458-1472)
      SC_EACH(M, r) {
        double temp = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) temp += L[C::oGu + r * NU + i] * kk[i];
        SC_ST(a.ky, t, M, r, msp ? (L[C::oRhat + r] + L[C::oY + r] * temp) / L[C::oSs + r] : clips(L[C::oRhat + r] + L[C::oY + r] * temp, L[C::oSs + r]));
        SC_ST(a.ks, t, M, r, (-L[C::oRp + r]) - temp);
      }
      SC_COLS(NX, c) {
        double kc[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) kc[i] = L[C::oKK + i * NX + c];
#pragma unroll 2
        for (int r = 0; r < M; ++r) {
          double s2 = 0.0;
#pragma unroll
          for (int i = 0; i < NU; ++i) s2 += L[C::oGu + r * NU + i] * kc[i];
          const int e = r * NX + c;
          const double gx = L[C::oGx + e];
          const double inner = gx + s2;
          SC_STU(a.Ky, t, M * NX, r * NX, msp ? L[C::oYS + r] * inner : dclamp(L[C::oYS + r] * inner, -kMaxRatioS, kMaxRatioS));
          SC_STU(a.Ks, t, M * NX, r * NX, (-gx) - s2);
        }
      }
      SC_TICK(6);
      // condensed terms into the Q blocks (:1488-1492)
      SC_EACH(NX, i) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += L[C::oGx + r * NX + i] * L[C::oSir + r];
        L[C::oQx + i] = L[C::oQx + i] + s1;
      }
      SC_COLS(NX, c) {
        double gxc[M], ys[M];
#pragma unroll
        for (int r = 0; r < M; ++r) { gxc[r] = L[C::oGx + r * NX + c]; ys[r] = L[C::oYS + r]; }
        sc_rows_t<NX, M>(L, [](int i, int r) { return C::oGx + r * NX + i; }, [&](int r, double x) { return (x * ys[r]) * gxc[r]; },
                         [&](int i, double s1) { L[C::oQxx + i * NX + c] = L[C::oQxx + i * NX + c] + s1; });
      }
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s1 += (L[C::oGu + r * NU + i] * L[C::oYS + r]) * L[C::oGu + r * NU + c];
        L[C::oQuu + e] = L[C::oQuu + e] + s1;
      }
      SC_EACH(NU, i) L[C::oQu + i] = L[C::oRu + i];
      if (msp) {   // msipddp_solver.cpp:1398 (see stacks.hip::sweep)
        SC_LOOP(NU * NX, e) {
          const int i = e / NX, c = e - i * NX;
          const int pi = (NU == 1) ? c : i, pc = (NU == 1) ? 0 : c;
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += (L[C::oGx + r * NX + pi] * L[C::oYS + r]) * L[C::oGu + r * NU + (pc < NU ? pc : 0)];
          L[C::oQux + e] = L[C::oQux + e] + s1;
        }
      } else {
        SC_LOOP(NU * NX, e) L[C::oQux + e] = L[C::oRx + e];
      }
    } else if (ip || lg) {
      // IPDDP: Q_uu = sym(Q_uu) + reg I, kept (:1084-1101).  LogDDP: factor sym(Q_uu + reg I), Q_uu itself untouched (:524-548)
      double qs[(NU * NU + 15) / 16];
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double p = L[C::oQuu + i * NU + c], q = L[C::oQuu + c * NU + i];
        if (lg) { if (i == c) { p += reg; q += reg; } qs[e_it] = 0.5 * (p + q); }
        else { double v = 0.5 * (p + q); if (i == c) v += reg; qs[e_it] = v; }
      }
      lds_sync();
      const bool caching = ms && !lg && a.QuuF != nullptr;   // MSIPDDP's per-step factor cache (msipddp_solver.cpp:1169-1185)
      const bool cached = caching && a.fvalid[(size_t)t * a.Bp + b] != 0;
      SC_EACH(NU * NU, e) {
        if (!lg) L[C::oQuu + e] = qs[e_it];
        L[C::oQr + e] = cached ? SC_LD(a.QuuF, t, NU * NU, e) : qs[e_it];
      }
      lds_sync();
      {
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQr + i];
        SCFactor<NU> f;
        if (!f.compute(Qr)) { if (caching && gl == 0) a.fvalid[(size_t)t * a.Bp + b] = 0; return false; }
        if (caching && !cached) {
          SC_EACH(NU * NU, e) SC_ST(a.QuuF, t, NU * NU, e, L[C::oQr + e]);
          if (gl == 0) a.fvalid[(size_t)t * a.Bp + b] = 1;
        }
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = L[C::oQu + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = L[C::oQux + i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = -col[i];
        }
      }
    } else {
      // CLDDP: PD test on Q_uu + reg I, BoxQP or dense inverse (clddp_solver.cpp:130-178); every lane of the group repeats it
      double Qr[NU * NU], Qu[NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQuu + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) { Qr[i * NU + i] += reg; Qu[i] = L[C::oQu + i]; }
      if (min_real_eig<NU>(Qr) <= 0) return false;
      if (a.lo) {
        double lb[NU], ub[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { const double ut = SC_LDV(a.U, t, NU, i); lb[i] = a.lo[i] - ut; ub[i] = a.up[i] - ut; kk[i] = SC_LDV(a.k, t, NU, i); }
        int free_[NU];
        LDLTd<NU> Hfree;
        const int stq = boxqp_solve<NU>(a0.opt, Qr, Qu, lb, ub, kk, free_, Hfree);
        if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
        int free_idx[NU]; int nf = 0;
        for (int i = 0; i < NU; ++i) if (free_[i]) free_idx[nf++] = i;
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = 0.0;
          if (nf > 0) {
            double col[NU];
            for (int i = 0; i < nf; ++i) col[i] = L[C::oQux + free_idx[i] * NX + c];
            Hfree.solve(col);
            for (int i = 0; i < nf; ++i) L[C::oKK + free_idx[i] * NX + c] = -col[i];
          }
        }
      } else {
        double H[NU * NU];
        inverse_pplu<NU>(Qr, H);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double s1 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) s1 += (-H[i * NU + j]) * Qu[j];
          kk[i] = s1;
        }
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) s2 += (-H[i * NU + j]) * L[C::oQux + j * NX + c];
            L[C::oKK + i * NX + c] = s2;
          }
        }
      }
    }
    lds_sync();
    SC_TICK(7);
    SC_EACH(NU, i) SC_ST(a.k, t, NU, i, kk[i]);
    SC_EACH(NU * NX, e) SC_ST(a.K, t, NU * NX, e, L[C::oKK + e]);
    {   // expected-decrease terms: the scalar chain every lane repeats
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) q += L[C::oQuu + i * NU + j] * kk[j];
        s0 += L[C::oQu + i] * kk[i]; s1 += kk[i] * q;
      }
      dV0 += s0; dV1 += 0.5 * s1;
#pragma unroll
      for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(L[C::oQu + i])); step_norm = dmax(step_norm, fabs(kk[i])); }
    }
    SC_LOOP(NX * NU, e) {   // K^T Q_uu
      const int i = e / NU, j = e - i * NU;
      double s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NU; ++k) s1 += L[C::oKK + k * NX + i] * L[C::oQuu + k * NU + j];
      L[C::oKtQ + e] = s1;
    }
    lds_sync();
    SC_TICK(8);
    // ---------------------------------------------------------------- value update
    double vxn[(NX + 15) / 16];
    SC_EACH(NX, i) {
      double p = 0.0, q = 0.0, r = 0.0;
      if (ip) {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKK + j * NX + i] * L[C::oQu + j]; q += L[C::oQux + j * NX + i] * kk[j]; r += L[C::oKtQ + i * NU + j] * kk[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKtQ + i * NU + j] * kk[j]; q += L[C::oQux + j * NX + i] * kk[j]; r += L[C::oKK + j * NX + i] * L[C::oQu + j]; }
      }
      vxn[i_it] = ((L[C::oQx + i] + p) + q) + r;
    }
    SC_COLS(NX, c) {
      double kc[NU], qc[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) { kc[j] = L[C::oKK + j * NX + c]; qc[j] = L[C::oQux + j * NX + c]; }
#pragma unroll 2
      for (int i = 0; i < NX; ++i) {
        double p = 0.0, q = 0.0, r = 0.0;
        if (ip) {
#pragma unroll
          for (int j = 0; j < NU; ++j) { p += L[C::oKK + j * NX + i] * qc[j]; q += L[C::oQux + j * NX + i] * kc[j]; r += L[C::oKtQ + i * NU + j] * kc[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < NU; ++j) { p += L[C::oKtQ + i * NU + j] * kc[j]; q += L[C::oQux + j * NX + i] * kc[j]; r += L[C::oKK + j * NX + i] * qc[j]; }
        }
        L[C::oVn + i * NX + c] = ((L[C::oQxx + i * NX + c] + p) + q) + r;
      }
    }
    SC_EACH(NX, i) { L[C::oVx + i] = vxn[i_it]; SC_ST(a.Vx, t, NX, i, vxn[i_it]); }
    lds_sync();
    SC_TICK(9);
    SC_EACH(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      const double v = 0.5 * (L[C::oVn + i * NX + c] + L[C::oVn + c * NX + i]);
      L[C::oVxx + e] = v; SC_ST(a.Vxx, t, NX * NX, e, v);
    }
    if (!ip && !lg) {
#pragma unroll
      for (int i = 0; i < NX; ++i) norm_Vx += fabs(L[C::oVx + i]);
    }
    lds_sync();
    SC_TICK(10);
  }
#ifdef SC_TIMING
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) for (int k = 0; k < 16; ++k) g_sc_times[blockIdx.x * 16 + k] = tk_acc[k];
#endif
  if constexpr (M > 0) {   // the lanes hold partial maxima over their constraint rows
    L[C::oRed + gl] = inf_pr; L[C::oRed + 16 + gl] = inf_comp;
    lds_sync();
    inf_pr = 0.0; inf_comp = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { inf_pr = dmax(inf_pr, L[C::oRed + i]); inf_comp = dmax(inf_comp, L[C::oRed + 16 + i]); }
    lds_sync();
  }
  if (!ip && !lg) {
    double sc = a.tau_min;
    sc = dmax(sc, norm_Vx / (double)(N * NX)) / sc;
    inf_du = inf_du / sc;
  }
  return true;
}


template <int NX, int NU, int M>
DEV bool sweep_coop(const StackArgs &a0, const int b, const int gl0, const typename SCOffFor<NX, NU, M>::type &vo, double *__restrict__ L, const double reg,
                    const double mu, double &dV0, double &dV1, double &inf_du, double &inf_pr, double &inf_comp, double &step_norm) {
  typedef SCfg<NX, NU, M> C;
  const StackArgs &a = a0;
  const int gl = gl0;
  const int N = a.N, bpo = a.Bp;
  // (a handle with path rows only takes the two path branches, stacks.hip: compile-time there, so that the value update carries one form)
  const bool lg = M > 0 ? false : a.branch == CDDP_HIP_STACKS_LOGDDP;
  const bool msp = a.branch == CDDP_HIP_STACKS_MSIPDDP_PATH;
  const bool ms = a.branch == CDDP_HIP_STACKS_MSIPDDP || msp;
  const bool ip = M > 0 ? true : (a.branch != CDDP_HIP_STACKS_CLDDP && !lg);
  SC_EACH(NX, i) L[C::oVx + i] = a.VxN[(size_t)i * a.Bp + b];
  if (ip || lg) {
    SC_LOOP(NX * NX, e) {
      const int i = e / NX, c = e - i * NX;
      L[C::oVxx + e] = 0.5 * (a.VxxN[(size_t)(i * NX + c) * a.Bp + b] + a.VxxN[(size_t)(c * NX + i) * a.Bp + b]);
    }
  } else {
    SC_LOOP(NX * NX, e) L[C::oVxx + e] = a.VxxN[(size_t)e * a.Bp + b];
  }
  lds_sync();
  SC_EACH(NX, i) SC_ST(a.Vx, N, NX, i, L[C::oVx + i]);
  SC_EACH(NX * NX, e) SC_ST(a.Vxx, N, NX * NX, e, L[C::oVxx + e]);
  dV0 = dV1 = 0.0; inf_du = inf_pr = inf_comp = step_norm = 0.0;
  double norm_Vx = 0.0;
  if (!ip && !lg) {
#pragma unroll
    for (int i = 0; i < NX; ++i) norm_Vx += fabs(L[C::oVx + i]);
  }
  SCRec<NX, NU, M> rec;
  rec.fetch(a, N - 1, gl, vo);
  rec.park(L, gl);
#ifdef SC_TIMING
  unsigned long long tk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_last = __builtin_readcyclecounter();
  const unsigned long long tk_rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock: slots 12 / 13 = start / end of the sweep, 14 = HW_ID | XCC_ID << 32
#endif
  const bool have_Fxx = a0.Fxx != nullptr;
  for (int t_ = N - 1; t_ >= 0; --t_) {
    SC_OPAQUE(t, bpo, t_);
    // The ~45 stack pointers of the argument block do not fit the scalar registers next to the step's buffer resources: kept across
    // the loop they were parked in VGPR lanes and came back through ~390 v_readlane per step.  Read through a pointer the optimiser
    // cannot see through, they are re-fetched from the kernel-argument segment (scalar cache) where a step needs them.
    sc_kargs_t kp = (sc_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const auto &a = *kp;
    // everything that hangs on the lane index (element and block indices, LDS addresses) is re-derived per step from an opaque copy:
    // as loop invariants they were hoisted out of the step loop and held well over a hundred registers across it
    int gl = gl0;
    asm volatile("" : "+v"(gl));
    // (the range is what turns "element e_it * 16 + gl exists" into a compile-time fact for all but the last e_it; only where it pays: with it
    //  the nx = 4 / m = 2 and nx = 3 / m = 5 instantiations of the cross-check shapes stop in the backend, "illegal VGPR to SGPR copy", ROCm 7.2)
    if constexpr (NX >= 12) gl &= 15;
    const int glt = gl;
    SC_TICK(15);
    // (the step's record is in LDS: parked group by group during the previous step, by the prologue for the first one)
    // w = V_x, or V_x + V_xx d_t under multiple shooting (V of step t + 1 is already in LDS)
    if (ms) {
      SC_EACH(NX, i) {
        double s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s1 += L[C::oVxx + i * NX + k] * SC_LDV(a.dfc, t, NX, k);
        L[C::oW + i] = L[C::oVx + i] + s1;
      }
    } else {
      SC_EACH(NX, i) L[C::oW + i] = L[C::oVx + i];
    }
    lds_sync();
    SC_TICK(0);
#if !SC_FETCH_LATE
    if (t > 0) rec.fetch(a, t - 1, gl, vo);   // in flight behind this step's arithmetic, parked at its end
#endif
    // ---------------------------------------------------------------- Q_x, Q_u, T1 = A^T V_xx, T2 = B^T V_xx
    {
      double qx[(NX + 15) / 16], qu[(NU + 15) / 16];
      SC_EACH(NX, i) {
        double q = L[C::oQx + i];
        if constexpr (M > 0) {
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += L[C::oGx + r * NX + i] * L[C::oY + r];
          q = q + s1;
        }
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += L[C::oA + k * NX + i] * L[C::oW + k];
        qx[i_it] = q + s2;
      }
      SC_EACH(NU, i) {
        double q = L[C::oQu + i];
        if constexpr (M > 0) {
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += L[C::oGu + r * NU + i] * L[C::oY + r];
          q = q + s1;
        }
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) s2 += L[C::oB + k * NU + i] * L[C::oW + k];
        qu[i_it] = q + s2;
      }
      SC_EACH(NX, i) L[C::oQx + i] = qx[i_it];
      SC_EACH(NU, i) L[C::oQu + i] = qu[i_it];
    }
    SC_TICK(1);
    sc_mm<NX, NX, NX, false>(L, glt, [](int i, int k) { return C::oA + k * NX + i; }, [](int k, int c) { return C::oVxx + k * NX + c; }, SCNoMid(),
                             [](int, int) { return 0.0; }, [&](int i, int c, double s1, double) { L[C::oT1 + i * NX + c] = s1; });
    sc_mm<NU, NX, NX, false>(L, glt, [](int i, int k) { return C::oB + k * NU + i; }, [](int k, int c) { return C::oVxx + k * NX + c; }, SCNoMid(),
                             [](int, int) { return 0.0; }, [&](int i, int c, double s1, double) { L[C::oT2 + i * NX + c] = s1; });
    lds_sync();
    SC_TICK(2);
    // ---------------------------------------------------------------- Q_xx += T1 A, Q_ux += T2 A, Q_uu += T2 B (+ tensor terms)
    sc_mm<NX, NX, NX, false>(L, glt, [](int i, int k) { return C::oT1 + i * NX + k; }, [](int k, int c) { return C::oA + k * NX + c; }, SCNoMid(),
                             [&](int i, int c) { return L[C::oQxx + i * NX + c]; }, [&](int i, int c, double s1, double old) { L[C::oQxx + i * NX + c] = old + s1; });
    sc_mm<NU, NX, NX, false>(L, glt, [](int i, int k) { return C::oT2 + i * NX + k; }, [](int k, int c) { return C::oA + k * NX + c; }, SCNoMid(),
                             [&](int i, int c) { return L[C::oQux + i * NX + c]; }, [&](int i, int c, double s1, double old) { L[C::oQux + i * NX + c] = old + s1; });
    sc_mm<NU, NU, NX, false>(L, glt, [](int i, int k) { return C::oT2 + i * NX + k; }, [](int k, int c) { return C::oB + k * NU + c; }, SCNoMid(),
                             [&](int i, int c) { return L[C::oQuu + i * NU + c]; }, [&](int i, int c, double s1, double old) { L[C::oQuu + i * NU + c] = old + s1; });
    if (have_Fxx) {   // second-order dynamics terms (use_ilqr = false), added to the finished sums in the same order; rolled: a cold path that must not shape the step's register allocation
      lds_sync();
      const __amdgpu_buffer_rsrc_t rFxx = SC_RS(a.Fxx, t, NX * NX * NX), rFux = SC_RS(a.Fux, t, NX * NU * NX), rFuu = SC_RS(a.Fuu, t, NX * NU * NU);
#pragma nounroll
      for (int e = gl; e < NX * NX; e += 16) {
        double q = L[C::oQxx + e];
#pragma nounroll
        for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * sc_ld(rFxx, (unsigned)(j * NX * NX + e) * SC_ES8 + vo.b8, 0u);
        L[C::oQxx + e] = q;
      }
#pragma nounroll
      for (int e = gl; e < NU * NX; e += 16) {
        double q = L[C::oQux + e];
#pragma nounroll
        for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * sc_ld(rFux, (unsigned)(j * NU * NX + e) * SC_ES8 + vo.b8, 0u);
        L[C::oQux + e] = q;
      }
#pragma nounroll
      for (int e = gl; e < NU * NU; e += 16) {
        double q = L[C::oQuu + e];
#pragma nounroll
        for (int j = 0; j < NX; ++j) q = q + L[C::oVx + j] * sc_ld(rFuu, (unsigned)(j * NU * NU + e) * SC_ES8 + vo.b8, 0u);
        L[C::oQuu + e] = q;
      }
    }
    lds_sync();
    SC_TICK(3);
    // ---------------------------------------------------------------- gains
    double kk[NU];
    if constexpr (M > 0) {
      const double s_floor = dmax(mu * 1e-3, kEpsSlackS);
      SC_EACH(M, r) {
        const double y = L[C::oY + r], s = L[C::oS + r], g = L[C::oGg + r];
        const double ssafe = msp ? s : dmax(s, s_floor);
        const double rp = g + s;
        const double rc = y * s - mu;
        const double rhat = y * rp - rc;
        L[C::oSs + r] = ssafe; L[C::oYS + r] = msp ? y / s : clipp(y, ssafe); L[C::oRp + r] = rp; L[C::oRhat + r] = rhat;
        L[C::oSir + r] = msp ? rhat / s : clips(rhat, ssafe);
        inf_pr = dmax(inf_pr, fabs(rp)); inf_comp = dmax(inf_comp, fabs(rc));
      }
      lds_sync();
      {   // R_u first (registers), then the two products; nothing below reads what they write
        double ru[(NU + 15) / 16];
        SC_EACH(NU, i) {
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += L[C::oGu + r * NU + i] * L[C::oSir + r];
          ru[i_it] = L[C::oQu + i] + s1;
        }
        SC_EACH(NU, i) L[C::oRu + i] = ru[i_it];
      }
      sc_mm<NU, NU, M, true>(L, glt, [](int i, int r) { return C::oGu + r * NU + i; }, [](int r, int c) { return C::oGu + r * NU + c; },
                             [](int r) { return C::oYS + r; }, [&](int i, int c) { return 0.5 * (L[C::oQuu + i * NU + c] + L[C::oQuu + c * NU + i]); },
                             [&](int i, int c, double s1, double sym) {
        double q = sym + s1;
        if (i == c) q += reg;
        L[C::oQr + i * NU + c] = q;
      });
      sc_mm<NU, NX, M, true>(L, glt, [](int i, int r) { return C::oGu + r * NU + i; }, [](int r, int c) { return C::oGx + r * NX + c; },
                             [](int r) { return C::oYS + r; }, [&](int i, int c) { return L[C::oQux + i * NX + c]; },
                             [&](int i, int c, double s2, double qux) { L[C::oRx + i * NX + c] = qux + s2; });
      lds_sync();
      SC_TICK(4);
      {
        double Qr[NU * NU], col[NU], colx[(NX + 15) / 16][NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQr + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = L[C::oRu + i];
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) colx[c_it][i] = L[C::oRx + i * NX + c];
        }
        SCFactor<NU> f;
        if (!f.compute(Qr)) return false;
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        SC_EACH(NX, c) f.solve(colx[c_it]);
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = -colx[c_it][i];
        }
      }
      lds_sync();
      SC_TICK(5);
      // slack / dual direction gains (:1458-1472): global stores only
      SC_EACH(M, r) {
        double temp = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) temp += L[C::oGu + r * NU + i] * kk[i];
        SC_ST(a.ky, t, M, r, msp ? (L[C::oRhat + r] + L[C::oY + r] * temp) / L[C::oSs + r] : clips(L[C::oRhat + r] + L[C::oY + r] * temp, L[C::oSs + r]));
        SC_ST(a.ks, t, M, r, (-L[C::oRp + r]) - temp);
      }
      {
        struct GY { double gx, ys; };
        const __amdgpu_buffer_rsrc_t rKy = SC_RS(a.Ky, t, M * NX), rKs = SC_RS(a.Ks, t, M * NX);   // (formed outside the lane-dependent write guard)
        sc_mm<M, NX, NU, false>(L, glt, [](int r, int i) { return C::oGu + r * NU + i; }, [](int i, int c) { return C::oKK + i * NX + c; }, SCNoMid(),
                                [&](int r, int c) { GY v; v.gx = L[C::oGx + r * NX + c]; v.ys = L[C::oYS + r]; return v; },
                                [&](int r, int c, double s2, GY v) {
          const double inner = v.gx + s2;
          const unsigned off = (unsigned)(r * NX + c) * SC_ES8 + vo.b8;
          sc_st(rKy, off, 0u, msp ? v.ys * inner : dclamp(v.ys * inner, -kMaxRatioS, kMaxRatioS));
          sc_st(rKs, off, 0u, (-v.gx) - s2);
        });
      }
      SC_TICK(6);
      // condensed terms into the Q blocks (:1488-1492)
      {
        double qx[(NX + 15) / 16], qu[(NU + 15) / 16];
        SC_EACH(NX, i) {
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += L[C::oGx + r * NX + i] * L[C::oSir + r];
          qx[i_it] = L[C::oQx + i] + s1;
        }
        SC_EACH(NU, i) qu[i_it] = L[C::oRu + i];
        SC_EACH(NX, i) L[C::oQx + i] = qx[i_it];
        SC_EACH(NU, i) L[C::oQu + i] = qu[i_it];
      }
      sc_mm<NX, NX, M, true>(L, glt, [](int i, int r) { return C::oGx + r * NX + i; }, [](int r, int c) { return C::oGx + r * NX + c; },
                             [](int r) { return C::oYS + r; }, [&](int i, int c) { return L[C::oQxx + i * NX + c]; },
                             [&](int i, int c, double s1, double old) { L[C::oQxx + i * NX + c] = old + s1; });
      sc_mm<NU, NU, M, true>(L, glt, [](int i, int r) { return C::oGu + r * NU + i; }, [](int r, int c) { return C::oGu + r * NU + c; },
                             [](int r) { return C::oYS + r; }, [&](int i, int c) { return L[C::oQuu + i * NU + c]; },
                             [&](int i, int c, double s1, double old) { L[C::oQuu + i * NU + c] = old + s1; });
      if (msp) {   // msipddp_solver.cpp:1398 (see stacks.hip::sweep)
        double qn[(NU * NX + 15) / 16];
        SC_EACH(NU * NX, e) {
          const int i = e / NX, c = e - i * NX;
          const int pi = (NU == 1) ? c : i, pc = (NU == 1) ? 0 : c;
          double s1 = 0.0;
#pragma unroll
          for (int r = 0; r < M; ++r) s1 += (L[C::oGx + r * NX + pi] * L[C::oYS + r]) * L[C::oGu + r * NU + (pc < NU ? pc : 0)];
          qn[e_it] = L[C::oQux + e] + s1;
        }
        SC_EACH(NU * NX, e) L[C::oQux + e] = qn[e_it];
      } else {
        double rx[(NU * NX + 15) / 16];
        SC_EACH(NU * NX, e) rx[e_it] = L[C::oRx + e];
        SC_EACH(NU * NX, e) L[C::oQux + e] = rx[e_it];
      }
    } else if (ip || lg) {
      // IPDDP: Q_uu = sym(Q_uu) + reg I, kept (:1084-1101).  LogDDP: factor sym(Q_uu + reg I), Q_uu itself untouched (:524-548)
      double qs[(NU * NU + 15) / 16];
      SC_EACH(NU * NU, e) {
        const int i = e / NU, c = e - i * NU;
        double p = L[C::oQuu + i * NU + c], q = L[C::oQuu + c * NU + i];
        if (lg) { if (i == c) { p += reg; q += reg; } qs[e_it] = 0.5 * (p + q); }
        else { double v = 0.5 * (p + q); if (i == c) v += reg; qs[e_it] = v; }
      }
      lds_sync();
      const bool caching = ms && !lg && a.QuuF != nullptr;   // MSIPDDP's per-step factor cache (msipddp_solver.cpp:1169-1185)
      const bool cached = caching && a.fvalid[(size_t)t * a.Bp + b] != 0;
      SC_EACH(NU * NU, e) {
        if (!lg) L[C::oQuu + e] = qs[e_it];
        L[C::oQr + e] = cached ? SC_LD(a.QuuF, t, NU * NU, e) : qs[e_it];
      }
      lds_sync();
      {
        double Qr[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQr + i];
        SCFactor<NU> f;
        if (!f.compute(Qr)) { if (caching && gl == 0) a.fvalid[(size_t)t * a.Bp + b] = 0; return false; }
        if (caching && !cached) {
          SC_EACH(NU * NU, e) SC_ST(a.QuuF, t, NU * NU, e, L[C::oQr + e]);
          if (gl == 0) a.fvalid[(size_t)t * a.Bp + b] = 1;
        }
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = L[C::oQu + i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = L[C::oQux + i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = -col[i];
        }
      }
    } else {
      // CLDDP: PD test on Q_uu + reg I, BoxQP or dense inverse (clddp_solver.cpp:130-178); every lane of the group repeats it
      double Qr[NU * NU], Qu[NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = L[C::oQuu + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) { Qr[i * NU + i] += reg; Qu[i] = L[C::oQu + i]; }
      if (min_real_eig<NU>(Qr) <= 0) return false;
      if (a.lo) {
        double lb[NU], ub[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { const double ut = SC_LDV(a.U, t, NU, i); lb[i] = a.lo[i] - ut; ub[i] = a.up[i] - ut; kk[i] = SC_LDV(a.k, t, NU, i); }
        int free_[NU];
        LDLTd<NU> Hfree;
        const int stq = boxqp_solve<NU>(a0.opt, Qr, Qu, lb, ub, kk, free_, Hfree);
        if (stq == BQ_HESSIAN_NOT_PD || stq == BQ_NO_DESCENT) return false;
        int free_idx[NU]; int nf = 0;
        for (int i = 0; i < NU; ++i) if (free_[i]) free_idx[nf++] = i;
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) L[C::oKK + i * NX + c] = 0.0;
          if (nf > 0) {
            double col[NU];
            for (int i = 0; i < nf; ++i) col[i] = L[C::oQux + free_idx[i] * NX + c];
            Hfree.solve(col);
            for (int i = 0; i < nf; ++i) L[C::oKK + free_idx[i] * NX + c] = -col[i];
          }
        }
      } else {
        double H[NU * NU];
        inverse_pplu<NU>(Qr, H);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double s1 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) s1 += (-H[i * NU + j]) * Qu[j];
          kk[i] = s1;
        }
        SC_EACH(NX, c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) s2 += (-H[i * NU + j]) * L[C::oQux + j * NX + c];
            L[C::oKK + i * NX + c] = s2;
          }
        }
      }
    }
    lds_sync();
    SC_TICK(7);
    SC_EACH(NU, i) SC_ST(a.k, t, NU, i, kk[i]);
    SC_EACH(NU * NX, e) SC_ST(a.K, t, NU * NX, e, L[C::oKK + e]);
    {   // expected-decrease terms: the scalar chain every lane repeats
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) q += L[C::oQuu + i * NU + j] * kk[j];
        s0 += L[C::oQu + i] * kk[i]; s1 += kk[i] * q;
      }
      dV0 += s0; dV1 += 0.5 * s1;
#pragma unroll
      for (int i = 0; i < NU; ++i) { inf_du = dmax(inf_du, fabs(L[C::oQu + i])); step_norm = dmax(step_norm, fabs(kk[i])); }
    }
    // K^T Q_uu
    sc_mm<NX, NU, NU, false>(L, glt, [](int i, int k) { return C::oKK + k * NX + i; }, [](int k, int j) { return C::oQuu + k * NU + j; }, SCNoMid(),
                             [](int, int) { return 0.0; }, [&](int i, int j, double s1, double) { L[C::oKtQ + i * NU + j] = s1; });
    lds_sync();
    SC_TICK(8);
#if SC_FETCH_LATE
    if (t > 0) rec.fetch(a, t - 1, gl, vo);   // behind V_n and the symmetrisation (~5 k clocks), parked at the end of the step
#endif
    // ---------------------------------------------------------------- value update
    double vxn[(NX + 15) / 16];
    SC_EACH(NX, i) {
      double p = 0.0, q = 0.0, r = 0.0;
      if (ip) {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKK + j * NX + i] * L[C::oQu + j]; q += L[C::oQux + j * NX + i] * kk[j]; r += L[C::oKtQ + i * NU + j] * kk[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += L[C::oKtQ + i * NU + j] * kk[j]; q += L[C::oQux + j * NX + i] * kk[j]; r += L[C::oKK + j * NX + i] * L[C::oQu + j]; }
      }
      vxn[i_it] = ((L[C::oQx + i] + p) + q) + r;
    }
    {   // V_n(i, c) = ((Q_xx + p) + q) + r, three NU-term sums per entry; the lane's block of entries, rows fetched one ahead
      typedef SCTile<NX, NX, true> T;
      constexpr int BR = T::BR, BC = T::BC;
      T tl; tl.init(glt);
      double kc[NU][BC], qc[NU][BC], vn[BR][BC];
#pragma unroll
      for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int q = 0; q < BC; ++q) { kc[j][q] = L[C::oKK + j * NX + tl.cj[q]]; qc[j][q] = L[C::oQux + j * NX + tl.cj[q]]; }
      struct Row { double kr[NU], qr[NU], tr[NU], qxx[BC]; };
      auto ldr = [&](const int r, Row &w) {
        const int i = tl.ri[r];
#pragma unroll
        for (int j = 0; j < NU; ++j) { w.kr[j] = L[C::oKK + j * NX + i]; w.qr[j] = L[C::oQux + j * NX + i]; w.tr[j] = L[C::oKtQ + i * NU + j]; }
#pragma unroll
        for (int q = 0; q < BC; ++q) w.qxx[q] = L[C::oQxx + i * NX + tl.cj[q]];
      };
      auto cmpr = [&](const int r, const Row &w) {
#pragma unroll
        for (int c = 0; c < BC; ++c) {
          double p = 0.0, q = 0.0, rr = 0.0;
          if (ip) {
#pragma unroll
            for (int j = 0; j < NU; ++j) { p += w.kr[j] * qc[j][c]; q += w.qr[j] * kc[j][c]; rr += w.tr[j] * kc[j][c]; }
          } else {
#pragma unroll
            for (int j = 0; j < NU; ++j) { p += w.tr[j] * kc[j][c]; q += w.qr[j] * kc[j][c]; rr += w.kr[j] * qc[j][c]; }
          }
          vn[r][c] = ((w.qxx[c] + p) + q) + rr;
        }
      };
      Row w0, w1;
      ldr(0, w0);
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        if (r + 1 < BR) { if ((r & 1) == 0) ldr(r + 1, w1); else ldr(r + 1, w0); }
        __builtin_amdgcn_sched_barrier(0);
        if ((r & 1) == 0) cmpr(r, w0); else cmpr(r, w1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int c = 0; c < BC; ++c)
          if (tl.vr[r] && tl.vc[c]) L[C::oVn + tl.ri[r] * NX + tl.cj[c]] = vn[r][c];
    }
    SC_EACH(NX, i) { L[C::oVx + i] = vxn[i_it]; SC_ST(a.Vx, t, NX, i, vxn[i_it]); }
    lds_sync();
    SC_TICK(9);
    if (t > 0) rec.park(L, gl);   // every area the record lands in is dead by now (V_n is in T1, the next reads are of V_n only)
    {
      double vs[(NX * NX + 15) / 16];
      SC_EACH(NX * NX, e) {
        const int i = e / NX, c = e - i * NX;
        vs[e_it] = 0.5 * (L[C::oVn + i * NX + c] + L[C::oVn + c * NX + i]);
      }
      SC_EACH(NX * NX, e) { L[C::oVxx + e] = vs[e_it]; SC_ST(a.Vxx, t, NX * NX, e, vs[e_it]); }
    }
    if (!ip && !lg) {
#pragma unroll
      for (int i = 0; i < NX; ++i) norm_Vx += fabs(L[C::oVx + i]);
    }
    lds_sync();
    SC_TICK(10);
  }
#ifdef SC_TIMING
  tk_acc[12] = tk_rt0; tk_acc[13] = __builtin_amdgcn_s_memrealtime();
  tk_acc[14] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) for (int k = 0; k < 16; ++k) g_sc_times[blockIdx.x * 16 + k] = tk_acc[k];
#endif
  if constexpr (M > 0) {   // the lanes hold partial maxima over their constraint rows
    L[C::oRed + gl] = inf_pr; L[C::oRed + 16 + gl] = inf_comp;
    lds_sync();
    inf_pr = 0.0; inf_comp = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { inf_pr = dmax(inf_pr, L[C::oRed + i]); inf_comp = dmax(inf_comp, L[C::oRed + 16 + i]); }
    lds_sync();
  }
  if (!ip && !lg) {
    double sc = a.tau_min;
    sc = dmax(sc, norm_Vx / (double)(N * NX)) / sc;
    inf_du = inf_du / sc;
  }
  return true;
}

// Round 6: the rollout's rows as TWO lane-parallel streams instead of five stack-by-stack ones.  A step needs a row of K_s and of K_y per
// constraint row and a row of K per control and of f_x (+ f_u) per state, each reduced against the same dx: lanes 0 - 7 take K_s rows and lanes
// 8 - 15 the K_y rows of the same constraint rows, and the NU + NX row tasks "K row | f_x row" are dealt over the sixteen lanes -- one
// instruction stream per element index j serves both halves (per-lane 64-bit pointers; a buffer resource would be per stack): 31 loads per
// step at nx = 12 where the stack-by-stack form issued 57 (and the same for the multiply-adds).  vmcnt tracks 63 loads: two steps in flight.
template <int NX, int NU, int M>
struct SCRoll2 {
  static constexpr int MM = M > 0 ? M : 1, RM = (MM + 7) / 8, RX = (NU + NX + 15) / 16;
  double cm[RM][NX], c0[RM], c1[RM];      // K_s | K_y row, k_s | k_y, s | y
  double rx[RX][NX], fu[RX][NU], kv[RX];  // K | f_x row, f_u row, k
};
template <int NX, int NU, int M>
struct SCRollPtr {   // the lane's element-0 addresses at step 0 and its strides per step, in doubles
  typedef SCRoll2<NX, NU, M> S;
  const gdouble *cm[S::RM], *c0[S::RM], *c1[S::RM], *rx[S::RX], *fu[S::RX], *kv[S::RX];
  unsigned rx_stride[S::RX];
  bool cvalid[S::RM], isK[S::RX], isF[S::RX];
  int row[S::RX];
  size_t es;   // doubles from one element to the next (Bp, or 4 in the tile-minor layout)
  template <class AT> DEV void init(const AT &a, int gl, int tl, int b) {
    const size_t Bp = (size_t)a.Bp;
    const bool t4 = a.t4 != 0;
    es = t4 ? 4 : Bp;
    auto at = [&](int E, int e) -> size_t { return t4 ? ((size_t)(b >> 2) * E + e) * 4 + (size_t)tl : (size_t)e * Bp + (size_t)b; };   // element e of step 0
    const int half = gl >> 3;
#pragma unroll
    for (int p = 0; p < S::RM; ++p) {
      const int r0 = (gl & 7) + 8 * p;
      cvalid[p] = r0 < M;
      const int r = r0 < M ? r0 : M - 1;
      cm[p] = (const gdouble *)(half ? a.Ky : a.Ks) + at(M * NX, r * NX);
      c0[p] = (const gdouble *)(half ? a.ky : a.ks) + at(M, r);
      c1[p] = (const gdouble *)(half ? a.y : a.s) + at(M, r);
    }
#pragma unroll
    for (int p = 0; p < S::RX; ++p) {
      const int id = gl + 16 * p;
      isK[p] = id < NU; isF[p] = id >= NU && id < NU + NX;
      row[p] = isK[p] ? id : (isF[p] ? id - NU : 0);
      rx[p] = isK[p] ? (const gdouble *)a.K + at(NU * NX, row[p] * NX) : (const gdouble *)a.fx + at(NX * NX, row[p] * NX);
      rx_stride[p] = (unsigned)((isK[p] ? NU * NX : NX * NX) * (int)Bp);
      fu[p] = (const gdouble *)a.fu + at(NX * NU, (isF[p] ? row[p] : 0) * NU);
      kv[p] = (const gdouble *)a.k + at(NU, isK[p] ? row[p] : 0);
    }
  }
  DEV void load(S &w, int t, int Bp_) const {
    const size_t Bp = (size_t)Bp_, tt = (size_t)t;
#pragma unroll
    for (int p = 0; p < S::RM; ++p) {
      const gdouble *q = cm[p] + tt * (size_t)(M * NX) * Bp;
#pragma unroll
      for (int j = 0; j < NX; ++j) w.cm[p][j] = q[(size_t)j * es];
      w.c0[p] = c0[p][tt * (size_t)M * Bp]; w.c1[p] = c1[p][tt * (size_t)M * Bp];
    }
#pragma unroll
    for (int p = 0; p < S::RX; ++p) {
      const gdouble *q = rx[p] + tt * (size_t)rx_stride[p];
#pragma unroll
      for (int j = 0; j < NX; ++j) w.rx[p][j] = q[(size_t)j * es];
      const gdouble *f = fu[p] + tt * (size_t)(NX * NU) * Bp;
#pragma unroll
      for (int j = 0; j < NU; ++j) w.fu[p][j] = f[(size_t)j * es];
      w.kv[p] = kv[p][tt * (size_t)NU * Bp];
    }
  }
};

template <int NX, int NU, int M>
__global__ __launch_bounds__(64) void k_stacks_backward_coop(StackArgs a) {
  typedef SCfg<NX, NU, M> C;
  extern __shared__ double sc_lds[];
  const int tl = threadIdx.x >> 4, gl = threadIdx.x & 15;
  const int b = sc_group((int)blockIdx.x) * 4 + tl;
  if (b >= a.B) return;
  double *L = sc_lds + tl * C::STRIDE;
  typename SCOffFor<NX, NU, M>::type vo;
  vo.init(gl, tl, b, a.Bp, a.t4);
  const double mu = a.mu ? a.mu[b] : 0.0;
  double reg = a.reg_in[b];
  double dV0 = 0, dV1 = 0, inf_du = 0, inf_pr = 0, inf_comp = 0, step_norm = 0;
  bool ok = false;
  for (;;) {
    if constexpr (NX >= SC_V1_NX) ok = sweep_coop_v1<NX, NU, M>(a, b, gl, vo, L, reg, mu, dV0, dV1, inf_du, inf_pr, inf_comp, step_norm);
    else ok = sweep_coop<NX, NU, M>(a, b, gl, vo, L, reg, mu, dV0, dV1, inf_du, inf_pr, inf_comp, step_norm);
    if (ok || !(a.reg_factor > 1.0)) break;
    reg = reg * a.reg_factor;
    if (!(reg > 0.0)) reg = (a.opt.reg_min_value > 0.0) ? a.opt.reg_min_value : a.reg_max;
    reg = dmin(reg, a.reg_max);
    if (reg >= a.reg_max) break;
    lds_sync();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // the next attempt reads what other lanes of the group stored (BoxQP warm start k, factor cache)
  }
  double apr = 1.0, adu = 1.0;
  if constexpr (M > 0) {
    if (ok && a.branch != CDDP_HIP_STACKS_MSIPDDP_PATH) {   // rolloutLinearPolicy from dx0 = 0, dS / dY, computeMaxStepSizes (ipddp_solver.cpp:1511-1532, 2939-2988)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // the gains were written through other lanes of this group
      const int N = a.N;
      const double tau = dmax(a.tau_min, 1.0 - mu);
      SC_EACH(NX, i) L[C::oW + i] = 0.0;
      lds_sync();
      const int bpo = a.Bp;
      typedef SCRoll2<NX, NU, M> RS;
      SCRollPtr<NX, NU, M> rp;
      rp.init(a, gl, tl, b);
      const int half = gl >> 3;
      double cap = 1.0;   // the lane's share of alpha_pr (lanes 0 - 7) or alpha_du (lanes 8 - 15)
      auto rstep = [&](const int t_, const RS &w) {   // one step on the rows in w
        SC_EACH(NX, i) SC_ST(a.dX, t_, NX, i, L[C::oW + i]);
        double dxr[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) dxr[j] = L[C::oW + j];
#pragma unroll
        for (int p = 0; p < RS::RM; ++p) {
          double pq = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) pq += w.cm[p][j] * dxr[j];
          const double d0 = w.c0[p] + pq;                                                  // dS_r on lanes 0 - 7, dY_r (clamped) on lanes 8 - 15
          const double dd = half == 0 ? d0 : dclamp(d0, -kMaxRatioS, kMaxRatioS);
          if (rp.cvalid[p] && dd < 0.0) cap = dmin(cap, -tau * w.c1[p] / dd);
        }
        double rs[RS::RX];
#pragma unroll
        for (int p = 0; p < RS::RX; ++p) {
          double sx = 0.0;
#pragma unroll
          for (int j = 0; j < NX; ++j) sx += w.rx[p][j] * dxr[j];
          rs[p] = sx;
          if (rp.isK[p]) L[C::oQu + rp.row[p]] = w.kv[p] + sx;
        }
        lds_sync();
        double dur[NU], dxn[RS::RX];
#pragma unroll
        for (int j = 0; j < NU; ++j) dur[j] = L[C::oQu + j];
#pragma unroll
        for (int p = 0; p < RS::RX; ++p) {
          double q = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) q += w.fu[p][j] * dur[j];
          dxn[p] = (rs[p] + q) + 0.0;
        }
#pragma unroll
        for (int p = 0; p < RS::RX; ++p) if (rp.isF[p]) L[C::oW + rp.row[p]] = dxn[p];
        lds_sync();
      };
      constexpr int NBUF = (NX <= 12) ? 3 : 2;   // rows of NBUF - 1 steps in flight (registers; vmcnt)
      RS st[NBUF];
#pragma unroll
      for (int u = 0; u < NBUF - 1; ++u) rp.load(st[u], u < N ? u : N - 1, bpo);
      int t0 = 0;
      for (; t0 + NBUF <= N; t0 += NBUF) {   // no branch around a load: the wait-count pass would merge "in flight" and "landed" at the join
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
          const int tn = t0 + u + NBUF - 1;
          rp.load(st[(u + NBUF - 1) % NBUF], tn < N ? tn : N - 1, bpo);
          rstep(t0 + u, st[u]);
        }
      }
      for (int u = 0; t0 + u < N; ++u) {   // tail: at most NBUF - 1 steps, their rows are in st[0 .. NBUF - 2]
        if (u == 0) rstep(t0, st[0]);
        else if (NBUF > 2 && u == 1) rstep(t0 + 1, st[1 % NBUF]);
      }
      apr = half == 0 ? cap : 1.0; adu = half == 0 ? 1.0 : cap;
      SC_EACH(NX, i) SC_ST(a.dX, N, NX, i, L[C::oW + i]);
      L[C::oRed + gl] = apr; L[C::oRed + 16 + gl] = adu;
      lds_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) { apr = dmin(apr, L[C::oRed + i]); adu = dmin(adu, L[C::oRed + 16 + i]); }
      apr = dclamp(apr, 0.0, 1.0); adu = dclamp(adu, 0.0, 1.0);
    }
  }
  if (gl == 0) {
    a.ok[b] = ok ? 1 : 0;
    a.dV[(size_t)0 * a.Bp + b] = dV0; a.dV[(size_t)1 * a.Bp + b] = dV1;
    a.scal[(size_t)0 * a.Bp + b] = reg; a.scal[(size_t)1 * a.Bp + b] = inf_du; a.scal[(size_t)2 * a.Bp + b] = inf_pr;
    a.scal[(size_t)3 * a.Bp + b] = inf_comp; a.scal[(size_t)4 * a.Bp + b] = step_norm;
    a.caps[(size_t)0 * a.Bp + b] = apr; a.caps[(size_t)1 * a.Bp + b] = adu;
  }
}

template <int NX, int NU, int M>
void launch_coop(const StackArgs &a, hipStream_t s) {
  typedef SCfg<NX, NU, M> C;
  constexpr size_t lds_bytes = (size_t)4 * C::STRIDE * sizeof(double);
  static_assert(lds_bytes <= 160 * 1024, "step state of four trajectories must fit the CU's LDS");
  static bool once = false;
  if (!once && lds_bytes > 64 * 1024) {
    hipFuncSetAttribute((const void *)k_stacks_backward_coop<NX, NU, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    once = true;
  }
  hipLaunchKernelGGL((k_stacks_backward_coop<NX, NU, M>), dim3(sc_grid(a.B)), dim3(64), lds_bytes, s, a);
}
