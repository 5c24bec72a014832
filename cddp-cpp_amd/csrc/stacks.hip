// Stack-fed Riccati backward sweep (cddp_hip_backward_stacks): host plugins evaluate arbitrary
// DynamicalSystem / Objective subclasses and hand over the (N x batch) derivative stacks; the GPU
// streams them once, one trajectory per lane, batch-minor so every wavefront load is a coalesced
// 512-B transaction.  This kernel is the pure HBM-streaming form of K2:
//   bytes read / trajectory  = 8 * (N*(nx^2 + nx*nu + nx + nu + nx^2 + nu^2 + nu*nx) + nx + nx^2)
//   bytes written            = 8 * (N*(nu*nx + nu + nx + nx^2) + nx + nx^2 + 2)
// Reference: ipddp_solver.cpp:1048-1118 (reg_in_value != 0) / clddp_solver.cpp:79-204 without bounds.
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "dev_linalg.hpp"
#include "../../include/cddp_hip.h"

using namespace cddp_dev;

namespace {

struct StackArgs {
  int B, Bp, N, reg_in_value;
  double reg;
  const double *fx, *fu, *lx, *lu, *lxx, *luu, *lux, *VxN, *VxxN;
  double *K, *k, *Vx, *Vxx, *dV;
  int *ok;
};

#define SI(t, E, e) ((((size_t)(t)) * (E) + (e)) * (size_t)a.Bp + (size_t)b)

template <int NX, int NU>
__global__ __launch_bounds__(64) void k_backward_stacks(StackArgs a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const int N = a.N;
  double Vx[NX], Vxx[NX * NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) Vx[i] = a.VxN[(size_t)i * a.Bp + b];
#pragma unroll
  for (int i = 0; i < NX * NX; ++i) Vxx[i] = a.VxxN[(size_t)i * a.Bp + b];
  if (a.reg_in_value) {   // V_xx = symmetrize(V_xx)  (ipddp_solver.cpp:992)
    double T[NX * NX];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) T[i] = Vxx[i];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (T[i * NX + c] + T[c * NX + i]);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) a.Vx[SI(N, NX, i)] = Vx[i];
#pragma unroll
  for (int i = 0; i < NX * NX; ++i) a.Vxx[SI(N, NX * NX, i)] = Vxx[i];
  double dV0 = 0.0, dV1 = 0.0;
  bool ok = true;
  for (int t = N - 1; t >= 0; --t) {
    double A[NX * NX], Bm[NX * NU], Qx[NX], Qu[NU], Qxx[NX * NX], Quu[NU * NU], Qux[NU * NX];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) A[i] = a.fx[SI(t, NX * NX, i)];
#pragma unroll
    for (int i = 0; i < NX * NU; ++i) Bm[i] = a.fu[SI(t, NX * NU, i)];
#pragma unroll
    for (int i = 0; i < NX; ++i) Qx[i] = a.lx[SI(t, NX, i)];
#pragma unroll
    for (int i = 0; i < NU; ++i) Qu[i] = a.lu[SI(t, NU, i)];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Qxx[i] = a.lxx[SI(t, NX * NX, i)];
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Quu[i] = a.luu[SI(t, NU * NU, i)];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) Qux[i] = a.lux[SI(t, NU * NX, i)];
#pragma unroll
    for (int i = 0; i < NX; ++i) { double s = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s += A[k * NX + i] * Vx[k];
      Qx[i] = Qx[i] + s; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s += Bm[k * NU + i] * Vx[k];
      Qu[i] = Qu[i] + s; }
    double T1[NX * NX], T2[NU * NX], P1[NX * NX], P2[NU * NX], P3[NU * NU];
    mm_tn<NX, NX, NX>(A, Vxx, T1);
    mm_tn<NU, NX, NX>(Bm, Vxx, T2);
    mm_nn<NX, NX, NX>(T1, A, P1);
    mm_nn<NU, NX, NX>(T2, A, P2);
    mm_nn<NU, NX, NU>(T2, Bm, P3);
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Qxx[i] = Qxx[i] + P1[i];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) Qux[i] = Qux[i] + P2[i];
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Quu[i] = Quu[i] + P3[i];
    double kk[NU], KK[NU * NX];
    if (a.reg_in_value) {
      double Qs[NU * NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int c = 0; c < NU; ++c) Qs[i * NU + c] = 0.5 * (Quu[i * NU + c] + Quu[c * NU + i]);
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Quu[i] = Qs[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Quu[i * NU + i] += a.reg;
      if (NU == 1) {
        kk[0] = -ldlt1_solve(Quu[0], Qu[0]);
#pragma unroll
        for (int c = 0; c < NX; ++c) KK[c] = -ldlt1_solve(Quu[0], Qux[c]);
      } else {
        LDLTd<NU> f;
        f.compute(Quu, NU);
        if (!f.ok) { ok = false; break; }
        double col[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) col[i] = Qu[i];
        f.solve(col);
#pragma unroll
        for (int i = 0; i < NU; ++i) kk[i] = -col[i];
        for (int c = 0; c < NX; ++c) {
#pragma unroll
          for (int i = 0; i < NU; ++i) col[i] = Qux[i * NX + c];
          f.solve(col);
#pragma unroll
          for (int i = 0; i < NU; ++i) KK[i * NX + c] = -col[i];
        }
      }
    } else {
      double Qr[NU * NU], H[NU * NU];
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Qr[i] = Quu[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) Qr[i * NU + i] += a.reg;
      if (min_real_eig<NU>(Qr) <= 0) { ok = false; break; }
      inverse_pplu<NU>(Qr, H);
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) s += (-H[i * NU + j]) * Qu[j];
        kk[i] = s;
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double s2 = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) s2 += (-H[i * NU + j]) * Qux[j * NX + c];
          KK[i * NX + c] = s2;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) a.k[SI(t, NU, i)] = kk[i];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) a.K[SI(t, NU * NX, i)] = KK[i];
    double Quuk[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) { double s = 0.0;
#pragma unroll
      for (int j = 0; j < NU; ++j) s += Quu[i * NU + j] * kk[j];
      Quuk[i] = s; }
    { double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) { s0 += Qu[i] * kk[i]; s1 += kk[i] * Quuk[i]; }
      dV0 += s0; dV1 += 0.5 * s1; }
    double KtQ[NX * NU];
    mm_tn<NX, NU, NU>(KK, Quu, KtQ);
    double Vn[NX * NX];
    if (a.reg_in_value) {   // IPDDP association order (ipddp_solver.cpp:1098-1101)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += KK[j * NX + i] * Qu[j]; q += Qux[j * NX + i] * kk[j]; r += KtQ[i * NU + j] * kk[j]; }
        Vx[i] = ((Qx[i] + p) + q) + r;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { p += KK[j * NX + i] * Qux[j * NX + c]; q += Qux[j * NX + i] * KK[j * NX + c]; r += KtQ[i * NU + j] * KK[j * NX + c]; }
          Vn[i * NX + c] = ((Qxx[i * NX + c] + p) + q) + r;
        }
    } else {                // CLDDP association order (clddp_solver.cpp:188-191)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) { p += KtQ[i * NU + j] * kk[j]; q += Qux[j * NX + i] * kk[j]; r += KK[j * NX + i] * Qu[j]; }
        Vx[i] = ((Qx[i] + p) + q) + r;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          double p = 0.0, q = 0.0, r = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) { p += KtQ[i * NU + j] * KK[j * NX + c]; q += Qux[j * NX + i] * KK[j * NX + c]; r += KK[j * NX + i] * Qux[j * NX + c]; }
          Vn[i * NX + c] = ((Qxx[i * NX + c] + p) + q) + r;
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) Vxx[i * NX + c] = 0.5 * (Vn[i * NX + c] + Vn[c * NX + i]);
#pragma unroll
    for (int i = 0; i < NX; ++i) a.Vx[SI(t, NX, i)] = Vx[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) a.Vxx[SI(t, NX * NX, i)] = Vxx[i];
  }
  a.dV[(size_t)0 * a.Bp + b] = dV0;
  a.dV[(size_t)1 * a.Bp + b] = dV1;
  a.ok[b] = ok ? 1 : 0;
}

thread_local std::string g_serr;
int sfail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_serr = buf;
  std::fprintf(stderr, "cddp_hip_backward_stacks: %s\n", buf);
  return code;
}

template <int NX, int NU>
void launch(const StackArgs &a, hipStream_t s) {
  hipLaunchKernelGGL((k_backward_stacks<NX, NU>), dim3((a.B + 63) / 64), dim3(64), 0, s, a);
}

void to_soa(const double *src, double *dst, int B, int Bp, int T, int E) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int e = 0; e < E; ++e) dst[((size_t)t * E + e) * Bp + b] = src[((size_t)b * T + t) * E + e];
}
void from_soa(const double *src, double *dst, int B, int Bp, int T, int E) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int e = 0; e < E; ++e) dst[((size_t)b * T + t) * E + e] = src[((size_t)t * E + e) * Bp + b];
}

}  // namespace

extern "C" int cddp_hip_backward_stacks(int device, int batch, int nx, int nu, int horizon, const double *fx,
                                        const double *fu, const double *lx, const double *lu, const double *lxx,
                                        const double *luu, const double *lux, const double *VxN, const double *VxxN,
                                        double reg, int reg_in_value, double *K, double *k, double *Vx, double *Vxx,
                                        double *dV, int32_t *ok, double *kernel_ms) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sfail(-20, "no HIP device available (no CPU fallback)");
  if (hipSetDevice(device) != hipSuccess) return sfail(-10, "hipSetDevice failed");
  void (*fn)(const StackArgs &, hipStream_t) = nullptr;
#define PICK(X, U) if (nx == X && nu == U) fn = &launch<X, U>;
  PICK(1, 1) PICK(2, 1) PICK(4, 1) PICK(3, 2) PICK(6, 3) PICK(12, 4) PICK(13, 4) PICK(14, 7)
#undef PICK
  if (!fn) return sfail(-4, "no stack-fed instantiation for nx=%d nu=%d", nx, nu);
  const int B = batch, Bp = (batch + 63) / 64 * 64, N = horizon;
  struct In { const double *src; int T, E; const double **dst; };
  StackArgs a;
  a.B = B; a.Bp = Bp; a.N = N; a.reg = reg; a.reg_in_value = reg_in_value;
  In ins[] = {{fx, N, nx * nx, &a.fx}, {fu, N, nx * nu, &a.fu}, {lx, N, nx, &a.lx}, {lu, N, nu, &a.lu},
              {lxx, N, nx * nx, &a.lxx}, {luu, N, nu * nu, &a.luu}, {lux, N, nu * nx, &a.lux},
              {VxN, 1, nx, &a.VxN}, {VxxN, 1, nx * nx, &a.VxxN}};
  std::vector<void *> allocs;
  auto cleanup = [&]() { for (void *q : allocs) hipFree(q); };
  std::vector<double> tmp;
  for (In &in : ins) {
    size_t n = (size_t)in.T * in.E * Bp;
    tmp.assign(n, 0.0);
    to_soa(in.src, tmp.data(), B, Bp, in.T, in.E);
    void *q = nullptr;
    if (hipMalloc(&q, n * 8) != hipSuccess) { cleanup(); return sfail(-10, "hipMalloc failed"); }
    allocs.push_back(q);
    hipMemcpy(q, tmp.data(), n * 8, hipMemcpyHostToDevice);
    *in.dst = (const double *)q;
  }
  struct Out { double **dev; double *host; int T, E; };
  Out outs[] = {{&a.K, K, N, nu * nx}, {&a.k, k, N, nu}, {&a.Vx, Vx, N + 1, nx}, {&a.Vxx, Vxx, N + 1, nx * nx}, {&a.dV, dV, 1, 2}};
  for (Out &o : outs) {
    size_t n = (size_t)o.T * o.E * Bp;
    void *q = nullptr;
    if (hipMalloc(&q, n * 8) != hipSuccess) { cleanup(); return sfail(-10, "hipMalloc failed"); }
    hipMemset(q, 0, n * 8);
    allocs.push_back(q);
    *o.dev = (double *)q;
  }
  { void *q = nullptr; if (hipMalloc(&q, Bp * 4) != hipSuccess) { cleanup(); return sfail(-10, "hipMalloc failed"); } allocs.push_back(q); a.ok = (int *)q; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  fn(a, nullptr);                       // warm-up launch (code object load)
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0, nullptr);
  for (int r = 0; r < reps; ++r) fn(a, nullptr);
  hipEventRecord(e1, nullptr);
  if (hipDeviceSynchronize() != hipSuccess) { cleanup(); return sfail(-10, "kernel failed: %s", hipGetErrorString(hipGetLastError())); }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  if (kernel_ms) *kernel_ms = ms / reps;
  hipEventDestroy(e0); hipEventDestroy(e1);
  for (Out &o : outs) {
    if (!o.host) continue;
    size_t n = (size_t)o.T * o.E * Bp;
    tmp.resize(n);
    hipMemcpy(tmp.data(), *o.dev, n * 8, hipMemcpyDeviceToHost);
    from_soa(tmp.data(), o.host, B, Bp, o.T, o.E);
  }
  if (ok) hipMemcpy(ok, a.ok, B * 4, hipMemcpyDeviceToHost);
  cleanup();
  return 0;
}
