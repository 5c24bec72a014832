// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_cartpole(std::vector<KernelSet> &v) {
  v.push_back(Launcher<CartPoleModel, ConList<>>::set("cartpole/none"));
  v.push_back(Launcher<CartPoleModel, ConList<CtrlBox<1>>>::set("cartpole/ctrlbox"));
}
}  // namespace cddp_dev

#ifdef CDDP_K4_TIMING
extern "C" int cddp_hip_debug_k4_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cddp_dev::g_k4_times), sizeof(unsigned long long) * (size_t)n);
}
#endif

#ifdef CDDP_ROLES_TIMING
extern "C" int cddp_hip_debug_roles_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cddp_dev::g_roles_times), sizeof(unsigned long long) * (size_t)n);
}
#endif
