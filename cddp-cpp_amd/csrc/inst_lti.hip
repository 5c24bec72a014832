// Explicit instantiation of the solver kernels for one plant (see launch.hpp).
#include "launch.hpp"
namespace cddp_dev {
void register_lti(std::vector<KernelSet> &v) {
  v.push_back(Launcher<LTIModel<1, 1>, ConList<>>::set("lti1x1/none"));
  v.push_back(Launcher<LTIModel<1, 1>, ConList<CtrlBox<1>>>::set("lti1x1/ctrlbox"));
  v.push_back(Launcher<LTIModel<1, 1>, ConList<Linear<1>>>::set("lti1x1/linear"));
  v.push_back(Launcher<LTIModel<2, 1>, ConList<>>::set("lti2x1/none"));
  v.push_back(Launcher<LTIModel<2, 1>, ConList<CtrlBox<1>>>::set("lti2x1/ctrlbox"));
}
}  // namespace cddp_dev
