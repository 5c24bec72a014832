// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_unicycle(std::vector<KernelSet> &v) {
  v.push_back(Launcher<UnicycleModel, ConList<>>::set("unicycle/none"));
  v.push_back(Launcher<UnicycleModel, ConList<CtrlBox<2>>>::set("unicycle/ctrlbox"));
  v.push_back(Launcher<UnicycleModel, ConList<CtrlBox<2>, Ball<2>>>::set("unicycle/ctrlbox+ball"));
}
}  // namespace cddp_dev
