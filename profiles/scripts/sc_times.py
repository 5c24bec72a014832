"""Experiment: cycles per section of the cooperative stack-fed sweep's step loop (library built with -DSC_TIMING for stacks.hip,
stacks_coop.hpp::SC_TICK).  usage: CDDP_HIP_LIB=.../libsc_time.so python profiles/scripts/sc_times.py nx nu m N batch [clddp]"""
import ctypes, importlib.util, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("pyapi", os.path.join(REPO, "cddp-cpp_amd", "pyapi.py"))
api = importlib.util.module_from_spec(spec); sys.modules["pyapi"] = api; spec.loader.exec_module(api)
nx, nu, m, N, B = [int(v) for v in sys.argv[1:6]]
clddp = len(sys.argv) > 6 and sys.argv[6] == "clddp"
rng = np.random.default_rng(1)
fx = np.tile(np.eye(nx), (B, N, 1, 1)) + 0.05 * rng.standard_normal((B, N, nx, nx)); fu = 0.1 * rng.standard_normal((B, N, nx, nu))
lx = rng.standard_normal((B, N, nx)); lu = rng.standard_normal((B, N, nu))
lxx = np.tile(np.eye(nx), (B, N, 1, 1)); luu = np.tile(np.eye(nu), (B, N, 1, 1)); lux = np.zeros((B, N, nu, nx))
VxN = rng.standard_normal((B, nx)); VxxN = np.tile(10.0 * np.eye(nx), (B, 1, 1))
mm = 0 if clddp else m
hs = api.HipStackSolver(B, nx, nu, mm, N)
hs.set_stacks(fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN)
mu = None
if mm:
    y = np.full((B, N, m), 0.5); sl = np.full((B, N, m), 0.4); g = -sl + 0.01 * rng.standard_normal((B, N, m))
    Gx = 0.1 * rng.standard_normal((B, N, m, nx)); Gu = 0.3 * rng.standard_normal((B, N, m, nu))
    hs.set_constraint_stacks(y, sl, g, Gx, Gu); mu = np.full(B, 0.1)
opt = api.default_options()
for _ in range(3):
    ok = hs.backward(api.STACKS_CLDDP if clddp else api.STACKS_IPDDP_PATH, opt, np.full(B, 1e-6), mu, retry=False); ms = hs.kernel_ms()
lib = ctypes.CDLL(api.HIP_LIB_PATH)
nb = min(4096, (B + 3) // 4)
buf = (ctypes.c_ulonglong * (nb * 16))()
assert lib.cddp_hip_debug_sc_times(buf, nb * 16) == 0
t = np.array(buf[:], dtype=np.float64).reshape(nb, 16)
t = t[t.sum(axis=1) > 0]
names = ["park + w", "Q_x, Q_u", "T1 = A^T V, T2 = B^T V", "Q_xx, Q_ux, Q_uu", "rows, Q_r, R_u, R_x", "factor + gain columns", "k_s k_y K_s K_y stores",
         "condensed terms into Q", "K k stores, dV, K^T Q_uu", "V_x, V_n", "V_xx sym + stores", "", "", "", "", "loop top"]
print("nx %d nu %d m %d N %d B %d %s: kernel %.3f ms, ok %d, form %d; clock ticks per STEP (median over %d workgroups)" % (nx, nu, mm, N, B, "CLDDP" if clddp else "IPDDP path", ms, int(ok.sum()), hs.sweep_form(), len(t)))
tot = 0.0
for i, nme in enumerate(names):
    if not nme: continue
    v = np.median(t[:, i]) / N; tot += v
    print("  %-28s %9.1f" % (nme, v))
print("  %-28s %9.1f  (kernel: %.1f us per step)" % ("sum", tot, ms * 1e3 / N))
if not t[:, 13].any():   # (the column-per-lane step of nx >= 13 carries the section timers only)
    hs.close(); sys.exit(0)
rt0, rt1 = t[:, 12] / 100.0, t[:, 13] / 100.0   # us
hw = t[:, 14].astype(np.uint64)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int); xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(int)
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
t00 = rt0.min()
print("  workgroup wall time (100 MHz clock): duration median %.1f us, min %.1f, max %.1f; starts: %d within 50 us of the first, latest start %.1f us, last end %.1f us" % (
    np.median(rt1 - rt0), (rt1 - rt0).min(), (rt1 - rt0).max(), int((rt0 - t00 < 50).sum()), (rt0 - t00).max(), (rt1 - t00).max()))
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key[rt0 - t00 < 50], return_counts=True)
print("  first-round workgroups per CU: %d CUs used, histogram of workgroups per CU %s; SIMD histogram %s" % (len(u), np.bincount(cnt).tolist(), np.bincount(simd).tolist()))
hs.close()
