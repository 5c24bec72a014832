// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_pendulum(std::vector<KernelSet> &v) {
  v.push_back(Launcher<PendulumModel, ConList<>>::set("pendulum/none"));
  v.push_back(Launcher<PendulumModel, ConList<CtrlBox<1>>>::set("pendulum/ctrlbox"));
}
}  // namespace cddp_dev

// Which sin / cos the reference plants of THIS build evaluate (dev_trig.hpp): 0 = device libm (product build),
// 1 = the shared branch-free routine (parity build, -DCDDP_TRIG_SHARED).  Lives in a translation unit that is compiled per variant.
extern "C" int cddp_hip_trig_shared(void) {
#ifdef CDDP_TRIG_SHARED
  return 1;
#else
  return 0;
#endif
}
