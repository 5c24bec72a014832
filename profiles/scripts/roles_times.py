"""Experiment: time stamps inside the LAST role-split sweep launch of a C2 solve (library built with -DCDDP_ROLES_TIMING for inst_cartpole.hip,
kernels_coop.hpp::ROLES_STAMP).  usage: CDDP_HIP_LIB=.../libroles_time.so python profiles/scripts/roles_times.py [batch] [max_iterations | 0] [cartpole | unicycle]"""
import ctypes, importlib.util, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("pyapi", os.path.join(REPO, "cddp-cpp_amd", "pyapi.py"))
api = importlib.util.module_from_spec(spec); sys.modules["pyapi"] = api; spec.loader.exec_module(api)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = sys.argv[3] if len(sys.argv) > 3 else "cartpole"
if wl == "unicycle":   # BASELINE config[2] (C3): N = 200, control box + ball
    p = api.unicycle_problem(api.SOLVER_IPDDP, 200, True)
    x0 = api.batch_x0(p, B, 20260928 + 1, [0.05, 0.05, 0.02]); U0 = api.batch_U0(p, B)
else:
    p = api.cartpole_problem(api.SOLVER_IPDDP, True)
    x0 = api.batch_x0(p, B, 20260928 + 1, [0.1, 0.3, 0.1, 0.1]); U0 = api.batch_U0(p, B)
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    p.options.max_iterations = int(sys.argv[2]); p._rebuild()
hs = api.HipBatchSolver(p, B)
hs.set_initial(x0, U0); hs.solve(); st = hs.solve()
lib = ctypes.CDLL(api.HIP_LIB_PATH)
nb = 128   # workgroups of ONE tile group (2048 trajectories / 16)
buf = (ctypes.c_ulonglong * (nb * 16))()
assert lib.cddp_hip_debug_roles_times(buf, nb * 16) == 0
t = np.array(buf[:], dtype=np.float64).reshape(nb, 16) / 100.0   # us (100 MHz)
t0 = t[:, 0:1]
names = {1: "rec: block 0 in the ring", 2: "rec: end of the backward recursion", 3: "rec: phase 2 published", 4: "rec: dX rollout done",
         5: "rec: helpers done (end)", 14: "rec: own share of the post blocks done", 8: "helper: start", 9: "helper: block 0 published", 10: "helper: last block of pass 0 published",
         11: "helper: phase 2 begins", 12: "helper: phase 2 done"}
print("B", B, "solve ms", round(st.solve_ms, 2), "(stamps relative to the recursion wave's start, us; median / p90 over", nb, "workgroups)")
for i in (8, 9, 1, 10, 2, 3, 11, 4, 14, 12, 5):
    v = t[:, i] - t0[:, 0]
    print("  %-42s %7.1f %7.1f" % (names[i], np.median(v), np.percentile(v, 90)))
print("  rec: time waiting for ring blocks (after block 0)  %7.1f %7.1f" % (np.median(t[:, 6]), np.percentile(t[:, 6], 90)))
hs.close()
