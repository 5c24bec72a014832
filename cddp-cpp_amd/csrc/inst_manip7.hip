// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_manip7(std::vector<KernelSet> &v) {
  v.push_back(Launcher<Manip7Model, ConList<CtrlBox<7>>>::set("manip7/ctrlbox"));
}
}  // namespace cddp_dev
