// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_quadrotor(std::vector<KernelSet> &v) {
  v.push_back(Launcher<QuadrotorModel, ConList<>>::set("quadrotor13/none"));
  v.push_back(Launcher<QuadrotorModel, ConList<CtrlBox<4>>>::set("quadrotor13/ctrlbox"));
}
}  // namespace cddp_dev
