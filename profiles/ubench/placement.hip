// Where do the two wavefronts of a 128-thread workgroup land?  704 workgroups (C2's rollout launch: 64 tiles x 11 step sizes) with the
// rollout's 36.9 KB of LDS each, every wave records HW_ID / XCC_ID and then spins ~50 us so that the whole grid is resident together.
// Output: per (XCC, SE, CU) the workgroups resident and, per SIMD, how many wave-0 / wave-1 roles it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void k_place(unsigned *out, int spin) {
  __shared__ double ring[8 * 9 * 64];
  const int w = threadIdx.x >> 6;
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_ID, all 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // XCC_ID
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 2 + w) * 2] = hw; out[(blockIdx.x * 2 + w) * 2 + 1] = xcc; }
  ring[threadIdx.x] = (double)hw;
  long long t0 = clock64();
  while (clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
  if (ring[(threadIdx.x + 1) & 127] == -1.0) out[0] = 0;
}
int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 704;
  unsigned *d; hipMalloc(&d, nb * 4 * sizeof(unsigned));
  hipLaunchKernelGGL(k_place, dim3(nb), dim3(128), 0, 0, d, 5000000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> cu;   // key: xcc, se, sh, cu -> per simd counts of role 0 / role 1
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 2; ++w) {
      unsigned hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xf;
      unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cuid;
      auto &v = cu[key]; if (v.empty()) v.assign(9, 0);
      v[simd * 2 + w]++; if (w == 0) v[8]++;
    }
  std::map<std::string, int> pattern;
  int first = 0;
  for (auto &kv : cu) {
    char buf[128]; const auto &v = kv.second;
    snprintf(buf, sizeof buf, "wg=%d simd0(P%d C%d) simd1(P%d C%d) simd2(P%d C%d) simd3(P%d C%d)", v[8], v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    pattern[buf]++;
    if (first++ < 6) printf("xcc %u se %u sh %u cu %2u: %s\n", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 0xf, buf);
  }
  printf("CUs used: %zu\n", cu.size());
  for (auto &p : pattern) printf("%4d CUs: %s\n", p.second, p.first.c_str());
  // block -> xcc order for the first blocks
  printf("block -> xcc:"); for (int b = 0; b < 24; ++b) printf(" %u", h[(b * 2) * 2 + 1] & 0xf); printf("\n");
  return 0;
}
