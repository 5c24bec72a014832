// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_cartpole(std::vector<KernelSet> &v) {
  v.push_back(Launcher<CartPoleModel, ConList<>>::set("cartpole/none"));
  v.push_back(Launcher<CartPoleModel, ConList<CtrlBox<1>>>::set("cartpole/ctrlbox"));
}
}  // namespace cddp_dev
