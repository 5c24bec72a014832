// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_quad12(std::vector<KernelSet> &v) {
  v.push_back(Launcher<Quad12Model, ConList<CtrlBox<4>>>::set("quad12/ctrlbox"));
}
}  // namespace cddp_dev
