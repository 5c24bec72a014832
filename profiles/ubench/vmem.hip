// micro-benchmark: cost of issuing K back-to-back coalesced row loads/stores from ONE wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W, int K>   // W = doubles per lane per instruction (1, 2), K = instructions per batch
__global__ void kload(const double *src, double *out, long long *cyc, int steps, size_t step_stride) {
  const size_t lane = threadIdx.x;
  const double *p = src + (size_t)blockIdx.x * 64 * W * K;   // this block's private rows per step
  double acc = 0;
  long long t_issue = 0, t_total = 0;
  for (int s = 0; s < steps; ++s) {
    const double *q = p + (size_t)s * step_stride;
    double v[K * W];
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (W == 1) v[k] = q[(size_t)k * 64 + lane];
      else { typedef double d2 __attribute__((ext_vector_type(2))); d2 r = *(const d2 *)(q + ((size_t)k * 64 + lane) * 2); v[2 * k] = r.x; v[2 * k + 1] = r.y; }
    }
    asm volatile("" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < K * W; ++k) acc += v[k];
    asm volatile("" : "+v"(acc));
    long long t2 = __builtin_readcyclecounter();
    t_issue += t1 - t0; t_total += t2 - t0;
  }
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) { cyc[2 * blockIdx.x] = t_issue; cyc[2 * blockIdx.x + 1] = t_total; }
}
template <int W, int K> void run(int blocks, int steps) {
  size_t step_stride = (size_t)blocks * 64 * W * K;
  size_t n = step_stride * steps;
  double *src, *out; long long *cyc;
  hipMalloc(&src, n * 8); hipMemset(src, 0, n * 8); hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 16);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((kload<W, K>), dim3(blocks), dim3(64), 0, 0, src, out, cyc, steps, step_stride); hipDeviceSynchronize(); }
  std::vector<long long> h(2 * blocks); hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
  double is = 0, tt = 0; for (int b = 0; b < blocks; ++b) { is += h[2 * b]; tt += h[2 * b + 1]; }
  printf("W=%d doubles/lane K=%2d instr blocks=%4d: issue %.0f cyc/batch (%.1f/instr), total %.0f cyc/batch, %.1f MB\n", W, K, blocks,
         is / blocks / steps, is / blocks / steps / K, tt / blocks / steps, n * 8 / 1e6);
  hipFree(src); hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {64, 704, 1408}) {
    run<1, 8>(blocks, 100); run<1, 16>(blocks, 100); run<1, 32>(blocks, 100); run<1, 48>(blocks, 100);
    run<2, 8>(blocks, 100); run<2, 16>(blocks, 100); run<2, 24>(blocks, 100);
  }
  return 0;
}
