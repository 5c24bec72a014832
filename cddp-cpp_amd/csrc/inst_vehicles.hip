// Explicit instantiation of the solver kernels for the two ground-vehicle plants, the HCW relative-motion plant and for the cone / thrust-magnitude constraint
// rows (constraint.hpp:626-1048) on the unicycle (see launch.hpp).  Constraint lists are in std::map (name) order:
// "ControlConstraint" < "SecondOrderConeConstraint"; the thrust rows under their reference names stand alone.
#include "launch.hpp"
namespace cddp_dev {
void register_vehicles(std::vector<KernelSet> &v) {
  v.push_back(Launcher<BicycleModel, ConList<>>::set("bicycle/none"));
  v.push_back(Launcher<BicycleModel, ConList<CtrlBox<2>>>::set("bicycle/ctrlbox"));
  v.push_back(Launcher<CarModel, ConList<>>::set("car/none"));
  v.push_back(Launcher<CarModel, ConList<CtrlBox<2>>>::set("car/ctrlbox"));
  v.push_back(Launcher<HCWModel, ConList<>>::set("hcw/none"));
  v.push_back(Launcher<HCWModel, ConList<CtrlBox<3>>>::set("hcw/ctrlbox"));
  v.push_back(Launcher<UnicycleModel, ConList<CtrlBox<2>, SecondOrderCone>>::set("unicycle/ctrlbox+soc"));
  v.push_back(Launcher<UnicycleModel, ConList<ThrustMagnitude<2, true>>>::set("unicycle/thrust"));
  v.push_back(Launcher<UnicycleModel, ConList<ThrustMagnitude<2, false>>>::set("unicycle/maxthrust"));
}
}  // namespace cddp_dev
