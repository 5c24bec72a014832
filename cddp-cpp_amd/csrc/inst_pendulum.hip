// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_pendulum(std::vector<KernelSet> &v) {
  v.push_back(Launcher<PendulumModel, ConList<>>::set("pendulum/none"));
  v.push_back(Launcher<PendulumModel, ConList<CtrlBox<1>>>::set("pendulum/ctrlbox"));
}
}  // namespace cddp_dev
