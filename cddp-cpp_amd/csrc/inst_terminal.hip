// Kernel sets with terminal constraints (TERM = true): terminal inequality (A_N x_N <= b_N) and the
// terminal-equality reduced-LQR branch.  Instantiated for the plants / layouts the reference's
// regression tests and BASELINE config 5 use.
#include "launch.hpp"
namespace cddp_dev {
void register_terminal(std::vector<KernelSet> &v) {
  v.push_back(Launcher<LTIModel<1, 1>, ConList<>, true>::set("lti1x1/none+terminal"));
  v.push_back(Launcher<LTIModel<1, 1>, ConList<Linear<1>>, true>::set("lti1x1/linear+terminal"));
  v.push_back(Launcher<LTIModel<1, 1>, ConList<CtrlBox<1>>, true>::set("lti1x1/ctrlbox+terminal"));
  v.push_back(Launcher<PendulumModel, ConList<CtrlBox<1>>, true>::set("pendulum/ctrlbox+terminal"));
  v.push_back(Launcher<ManipulatorModel, ConList<CtrlBox<3>>, true>::set("manipulator3/ctrlbox+terminal"));
  v.push_back(Launcher<Manip7Model, ConList<CtrlBox<7>>, true>::set("manip7/ctrlbox+terminal"));
}
}  // namespace cddp_dev
