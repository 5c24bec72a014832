// micro-benchmark: dependent / independent f64 VALU issue cost on one wave per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP, int CH>
__global__ void k(double *out, long long *cyc, int iters, double a0, double b0) {
  double v[CH];
  for (int c = 0; c < CH; ++c) v[c] = a0 + c + threadIdx.x * 1e-3;
  double b = b0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[c]) : "v"(b));
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
        if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
        if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[c]));
        if (OP == 4) { float f; asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(f) : "v"((float)v[c])); v[c] = f; }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int c = 0; c < CH; ++c) s += v[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int CH> void run(const char *name, int blocks) {
  double *out; long long *cyc; hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
  int iters = 2000;
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0, 1.0000001);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0, 1.0000001);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * CH;
  printf("%-10s chains=%d blocks=%4d: %.2f memtime-ticks/instr, %.2f ns/instr (wall)\n", name, CH, blocks, h[0] / n, ms * 1e6 / n);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1, 1024, 2048}) {
    run<0, 1>("fma_f64", blocks); run<0, 2>("fma_f64", blocks); run<0, 4>("fma_f64", blocks);
    run<1, 1>("add_f64", blocks); run<1, 4>("add_f64", blocks);
    run<2, 1>("mul_f64", blocks); run<2, 4>("mul_f64", blocks);
    run<3, 1>("rcp_f64", blocks); run<3, 4>("rcp_f64", blocks);
  }
  return 0;
}
