// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_unicycle(std::vector<KernelSet> &v) {
  v.push_back(Launcher<UnicycleModel, ConList<>>::set("unicycle/none"));
  v.push_back(Launcher<UnicycleModel, ConList<CtrlBox<2>>>::set("unicycle/ctrlbox"));
  v.push_back(Launcher<UnicycleModel, ConList<CtrlBox<2>, Ball<2>>>::set("unicycle/ctrlbox+ball"));
}
}  // namespace cddp_dev

#ifdef CDDP_ROLES_TIMING
extern "C" int cddp_hip_debug_roles_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cddp_dev::g_roles_times), sizeof(unsigned long long) * (size_t)n);
}
#endif
