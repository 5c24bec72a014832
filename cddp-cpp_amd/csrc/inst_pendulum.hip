// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_pendulum(std::vector<KernelSet> &v) {
  v.push_back(Launcher<PendulumModel, ConList<>>::set("pendulum/none"));
  v.push_back(Launcher<PendulumModel, ConList<CtrlBox<1>>>::set("pendulum/ctrlbox"));
}
}  // namespace cddp_dev

// Which sin / cos / log / pow THIS build evaluates (dev_trig.hpp): 1 = the shared straight-line routines (the only build the
// Makefile produces since round 4); 0 = the device libm (an experiment build made by hand without -DCDDP_TRIG_SHARED).
extern "C" int cddp_hip_trig_shared(void) {
#ifdef CDDP_TRIG_SHARED
  return 1;
#else
  return 0;
#endif
}
