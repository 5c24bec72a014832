// Layouts with a StateConstraint (BoxConstraint<State>, constraint.hpp:144-251) next to the control box: the
// constraint rows read x, so the condensation carries G_x blocks (Cons::HAS_X).  Constraint objects are ordered by
// name as std::map iterates them: "ControlConstraint" < "StateConstraint".
#include "launch.hpp"
namespace cddp_dev {
void register_statebox(std::vector<KernelSet> &v) {
  v.push_back(Launcher<PendulumModel, ConList<CtrlBox<1>, StateBox<2>>>::set("pendulum/ctrlbox+statebox"));
  v.push_back(Launcher<CartPoleModel, ConList<CtrlBox<1>, StateBox<4>>>::set("cartpole/ctrlbox+statebox"));
  v.push_back(Launcher<UnicycleModel, ConList<CtrlBox<2>, StateBox<3>>>::set("unicycle/ctrlbox+statebox"));
  v.push_back(Launcher<LTIModel<2, 1>, ConList<StateBox<2>>>::set("lti2x1/statebox"));
}
}  // namespace cddp_dev
