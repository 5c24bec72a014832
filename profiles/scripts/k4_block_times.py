"""Experiment: per-wave duration of the LAST rollout launch of a C2 solve (library built with -DCDDP_K4_TIMING, see kernels_lean.hpp).
usage: CDDP_HIP_LIB=.../libcddp_hip_time.so python profiles/scripts/k4_block_times.py [batch] [max_iterations]"""
import ctypes, importlib.util, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("pyapi", os.path.join(REPO, "cddp-cpp_amd", "pyapi.py"))
api = importlib.util.module_from_spec(spec); sys.modules["pyapi"] = api; spec.loader.exec_module(api)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = api.cartpole_problem(api.SOLVER_IPDDP, True)
if len(sys.argv) > 2:
    p.options.max_iterations = int(sys.argv[2]); p._rebuild()
x0 = api.batch_x0(p, B, 20260928 + 1, [0.1, 0.3, 0.1, 0.1]); U0 = api.batch_U0(p, B)
hs = api.HipBatchSolver(p, B)
hs.set_initial(x0, U0); hs.solve(); st = hs.solve()
lib = ctypes.CDLL(api.HIP_LIB_PATH)
nb = ((B + 63) // 64) * 11
buf = (ctypes.c_ulonglong * (nb * 4))()
assert lib.cddp_hip_debug_k4_times(buf, nb * 4) == 0
t = np.array(buf[:], dtype=np.float64).reshape(nb, 2, 2)   # [block][role][time (100 MHz ticks), steps]
us = t[:, :, 0] / 100.0
tiles = (B + 63) // 64
print("B", B, "blocks", nb, "solve ms", round(st.solve_ms, 2))
for role, name in ((0, "producer"), (1, "consumer")):
    u = us[:, role]
    print(name, "us: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (u.min(), np.percentile(u, 10), np.median(u), np.percentile(u, 90), u.max()))
print("per alpha (median producer us | median consumer steps):")
for a in range(11):
    blk = slice(a * tiles, (a + 1) * tiles)
    print("  alpha %2d: %6.0f us | steps %5.0f | max %6.0f us" % (a, np.median(us[blk, 0]), np.median(t[blk, 1, 1]), us[blk, 0].max()))
hs.close()
