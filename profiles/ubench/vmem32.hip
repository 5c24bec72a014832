// micro-benchmark / counter calibration: the access pattern of the G = 16 cooperative sweeps (kernels_coop.hpp, kernels_te.hpp).
// A single-wave workgroup holds 4 trajectories x 16 lanes; every load instruction reads 16 rows x 32 B (4 adjacent trajectories of
// 16 different 512-B rows of the wave-tiled stacks).  The 16 workgroups of one 64-trajectory tile either are consecutive block
// indices (map 0: round-robin over the XCDs, every 128-B line is fetched by up to four L2s) or share an XCD (map 1:
// kernels_coop.hpp::coop_group).  Known bytes: blocks * steps * K * 512.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./vmem32      (then --pmc WRITE_SIZE)
// to read the counter's scale on THIS pattern (MI355X_MICROARCH.md: FETCH_SIZE is calibrated on wide coalesced streams only).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int MAP>
__global__ void kload32(const double *src, double *out, int steps, int NB, int E) {
  // wave-tiled stack: element e of step t of trajectory b at (((t * NB + b / 64) * E + e) * 64 + b % 64)
  const int lane = threadIdx.x, q = lane & 15, tl = lane >> 4;
  int bid = blockIdx.x;
  if (MAP) { const int sup = bid / 128, r = bid % 128; bid = (sup * 8 + (r & 7)) * 16 + (r >> 3); }
  const int b = bid * 4 + tl;
  double acc = 0;
  for (int s = 0; s < steps; ++s) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = src[(((size_t)s * NB + (b >> 6)) * E + (q + 16 * k)) * 64 + (b & 63)];
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k];
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc;
}
template <int K, int MAP> void run(int B, int steps) {
  const int NB = B / 64, E = 16 * K, blocks = B / 4;
  const size_t n = (size_t)steps * NB * E * 64;
  double *src, *out;
  hipMalloc(&src, n * 8); hipMemset(src, 0, n * 8); hipMalloc(&out, (size_t)blocks * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kload32<K, MAP>), dim3(blocks), dim3(64), 0, 0, src, out, steps, NB, E);
    hipEventRecord(e1); hipDeviceSynchronize();
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("kload32<K=%d, MAP=%d> B=%d steps=%d: %.1f MB read once, %.3f ms, %.0f GB/s\n", K, MAP, B, steps, n * 8 / 1e6, ms, n * 8 / 1e6 / ms);
  hipFree(src); hipFree(out);
}
int main() {
  run<9, 0>(2048, 400); run<9, 1>(2048, 400);     // ~ the C4 sweep's A_t record (144 doubles per trajectory and step)
  run<12, 0>(4096, 150); run<12, 1>(4096, 150);   // ~ the C5 sweep's record
  return 0;
}
