#!/bin/bash
# Reference pin, step 1 of 2 (test infrastructure): build cddp-cpp's OWN solver core + plants from the sources
# where they lie under /root/reference (nothing is copied) with plain g++, and link them with dump_traces.cpp into
# oracle/_ref/dump_traces.  The reference needs two header-only third-party libraries that this image does NOT hold and that this
# script does NOT stand in for:
#     Eigen     3.4.0   (CMakeLists.txt:65-97: find_package(Eigen3 3.4) / FetchContent GIT_TAG 3.4.0)
#     autodiff  v1.1.2  (CMakeLists.txt:116-125: FetchContent GIT_TAG v1.1.2)
# Point the two environment variables at checkouts of exactly those tags:
#     EIGEN3_INCLUDE_DIR=/path/to/eigen-3.4.0            (the directory that contains Eigen/Dense)
#     AUTODIFF_INCLUDE_DIR=/path/to/autodiff-1.1.2       (the directory that contains autodiff/forward/dual.hpp)
# and run   oracle/ref_pin/build_ref.sh && python oracle/ref_pin/compare_traces.py
# Without them the script stops here -- there is no fallback, and no oracle/_ref in this image ("parity unpinned").
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
REF="${CDDP_REFERENCE_DIR:-/root/reference}"
OUT="$REPO/oracle/_ref"
fail() { echo "build_ref.sh: $*" >&2; exit 2; }
[ -d "$REF/src/cddp_core" ] || fail "reference sources not found under $REF (set CDDP_REFERENCE_DIR)"
[ -n "${EIGEN3_INCLUDE_DIR:-}" ] || fail "EIGEN3_INCLUDE_DIR is not set (needs an Eigen 3.4.0 checkout; not vendored, not installed, no stand-in)"
[ -n "${AUTODIFF_INCLUDE_DIR:-}" ] || fail "AUTODIFF_INCLUDE_DIR is not set (needs an autodiff v1.1.2 checkout; not vendored, not installed, no stand-in)"
[ -f "$EIGEN3_INCLUDE_DIR/Eigen/Dense" ] || fail "$EIGEN3_INCLUDE_DIR/Eigen/Dense does not exist"
[ -f "$AUTODIFF_INCLUDE_DIR/autodiff/forward/dual.hpp" ] || fail "$AUTODIFF_INCLUDE_DIR/autodiff/forward/dual.hpp does not exist"
M="$EIGEN3_INCLUDE_DIR/Eigen/src/Core/util/Macros.h"
ver="$(sed -n 's/^#define EIGEN_WORLD_VERSION \([0-9]*\)/\1/p' "$M").$(sed -n 's/^#define EIGEN_MAJOR_VERSION \([0-9]*\)/\1/p' "$M").$(sed -n 's/^#define EIGEN_MINOR_VERSION \([0-9]*\)/\1/p' "$M")"
[ "$ver" = "3.4.0" ] || fail "Eigen $ver found, the reference pins 3.4.0 (LDLT / PartialPivLU / JacobiSVD semantics steer solver decisions)"
mkdir -p "$OUT/obj"
# the CMake Release configuration of the reference: C++17, -O3 -DNDEBUG, no -march (plain x86-64: no FMA contraction -- the
# oracle and the HIP library are built with -ffp-contract=off for the same reason)
CXXFLAGS="-std=c++17 -O3 -DNDEBUG -ffp-contract=off -I$REF/include/cddp-cpp -I$REF/include -I$REF/src/cddp_core -I$EIGEN3_INCLUDE_DIR -I$AUTODIFF_INCLUDE_DIR"
SRCS=()
for f in "$REF"/src/cddp_core/*.cpp; do SRCS+=("$f"); done
for m in pendulum cartpole unicycle quadrotor manipulator lti_system car bicycle spacecraft_linear; do SRCS+=("$REF/src/dynamics_model/$m.cpp"); done
OBJS=()
for f in "${SRCS[@]}"; do
  o="$OUT/obj/$(basename "${f%.cpp}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then echo "g++ $(basename "$f")"; g++ $CXXFLAGS -c "$f" -o "$o"; fi
  OBJS+=("$o")
done
g++ $CXXFLAGS "$HERE/dump_traces.cpp" "${OBJS[@]}" -o "$OUT/dump_traces" -pthread
echo "built $OUT/dump_traces  (next: python oracle/ref_pin/compare_traces.py)"
# Step 1b (round 5): the reference-side ADAPTER (integration/hip_batch_solver.cpp: ISolverAlgorithm over the C-ABI, registered as "IPDDP" /
# "CLDDP" / "LogDDP" / "MSIPDDP") compiled against the same real headers and linked with the reference objects and libcddp_hip.so, plus its
# registry / drop-in checks (integration/test_hip_registry.cpp: test_cddp_core.cpp:316-411, 463-483 restated; reference solver vs GPU solver
# iteration counts on the pendulum / cart-pole examples).  Run  LD_LIBRARY_PATH=cddp-cpp_amd/lib oracle/_ref/test_hip_registry  on an MI355X box.
HIPLIB="$REPO/cddp-cpp_amd/lib"
[ -f "$HIPLIB/libcddp_hip.so" ] || fail "$HIPLIB/libcddp_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
g++ $CXXFLAGS -I"$REPO/include" -I"$REPO/integration" -c "$REPO/integration/hip_batch_solver.cpp" -o "$OUT/obj/hip_batch_solver.o"
g++ $CXXFLAGS -I"$REPO/include" -I"$REPO/integration" "$REPO/integration/test_hip_registry.cpp" "$OUT/obj/hip_batch_solver.o" "${OBJS[@]}" \
    -L"$HIPLIB" -lcddp_hip -Wl,-rpath,"$HIPLIB" -o "$OUT/test_hip_registry" -pthread
echo "built $OUT/test_hip_registry  (run it on a box with an MI355X: exit code 0 = the registered GPU solvers reproduce the reference's)"
