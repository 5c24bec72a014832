// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_manipulator(std::vector<KernelSet> &v) {
  v.push_back(Launcher<ManipulatorModel, ConList<>>::set("manipulator3/none"));
  v.push_back(Launcher<ManipulatorModel, ConList<CtrlBox<3>>>::set("manipulator3/ctrlbox"));
}
}  // namespace cddp_dev
