// v_mfma_f64_16x16x4_f64 operand / result layouts on gfx950, and the register-chaining identities the MFMA sweep
// (cddp-cpp_amd/csrc/kernels_mfma.hpp) relies on.  hipcc --offload-arch=gfx950 -O2 mfma_f64.hip -o mfma_f64 && ./mfma_f64
//   A operand (16 x 4):  lane l holds A[i = l % 16][k = l / 16]
//   B operand (4 x 16):  lane l holds B[k = l / 16][j = l % 16]
//   C / D     (16 x 16): lane l, register v holds D[row = l / 16 + 4 v][col = l % 16]
// Chaining: a 16x16 result X in D layout is, register by register, the B operand of k-step v when the k-steps are
// taken over the interleaved index sets {g + 4 v : g = 0..3}; and for symmetric X the same registers are also the A
// operand (lane (g, c) needs X[c][g + 4 v] = X[g + 4 v][c]).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_test(const double *A, const double *B, const double *S, double *out1, double *out2, double *out3, long long *cyc) {
  const int l = threadIdx.x, g = l >> 4, c = l & 15;
  // product 1: D1 = A * B with the interleaved k-sets: k-step v uses k = g + 4 v
  d4 acc = {0, 0, 0, 0};
  for (int v = 0; v < 4; ++v) {
    const double a = A[c * 16 + (g + 4 * v)];        // A-op: A[i = c][k = g + 4v]
    const double b = B[(g + 4 * v) * 16 + c];        // B-op: B[k = g + 4v][j = c]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int v = 0; v < 4; ++v) out1[(g + 4 * v) * 16 + c] = acc[v];
  // product 2: D2 = A^T * D1, D1 straight from the accumulator registers (B operand of k-step v = acc[v]);
  // A^T as A-op: lane (i = c, g) needs A^T[c][g + 4v] = A[g + 4v][c]
  d4 acc2 = {0, 0, 0, 0};
  for (int v = 0; v < 4; ++v) {
    const double a = A[(g + 4 * v) * 16 + c];
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[v], acc2, 0, 0, 0);
  }
  for (int v = 0; v < 4; ++v) out2[(g + 4 * v) * 16 + c] = acc2[v];
  // product 3: D3 = S * B with symmetric S held in D layout (sreg[v] = S[g + 4v][c]) used directly as the A operand
  double sreg[4];
  for (int v = 0; v < 4; ++v) sreg[v] = S[(g + 4 * v) * 16 + c];
  d4 acc3 = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int rep = 0; rep < 64; ++rep)
    for (int v = 0; v < 4; ++v) acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sreg[v], B[(g + 4 * v) * 16 + c], acc3, 0, 0, 0);
  long long t1 = clock64();
  for (int v = 0; v < 4; ++v) out3[(g + 4 * v) * 16 + c] = acc3[v] / 64.0;
  if (l == 0) *cyc = (t1 - t0) / 256;   // dependent-chain cycles per MFMA
}

int main() {
  std::vector<double> A(256), B(256), S(256), r1(256), r2(256), r3(256);
  srand(7);
  for (int i = 0; i < 256; ++i) { A[i] = rand() / (double)RAND_MAX - 0.5; B[i] = rand() / (double)RAND_MAX - 0.3; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) S[i * 16 + j] = S[j * 16 + i] = rand() / (double)RAND_MAX;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 16 + j]; r1[i * 16 + j] = s; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += A[k * 16 + i] * r1[k * 16 + j]; r2[i * 16 + j] = s; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += S[i * 16 + k] * B[k * 16 + j]; r3[i * 16 + j] = s; }
  double *dA, *dB, *dS, *o1, *o2, *o3; long long *dc;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dS, 2048); hipMalloc(&o1, 2048); hipMalloc(&o2, 2048); hipMalloc(&o3, 2048); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dS, S.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dS, o1, o2, o3, dc);
  std::vector<double> h1(256), h2(256), h3(256); long long cyc = 0;
  hipMemcpy(h1.data(), o1, 2048, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, 2048, hipMemcpyDeviceToHost); hipMemcpy(h3.data(), o3, 2048, hipMemcpyDeviceToHost);
  hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < 256; ++i) { e1 = fmax(e1, fabs(h1[i] - r1[i])); e2 = fmax(e2, fabs(h2[i] - r2[i])); e3 = fmax(e3, fabs(h3[i] - r3[i])); }
  printf("A*B max err %.3e | A^T*(A*B) chained from registers max err %.3e | S*B (symmetric S as A-op from D layout) max err %.3e | %lld cycles per dependent MFMA\n", e1, e2, e3, cyc);
  return (e1 < 1e-12 && e2 < 1e-12 && e3 < 1e-12) ? 0 : 1;
}
