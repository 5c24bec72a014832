# whole C5 solves + single sweeps of every terminal-equality plant: dump gains, value gradients, trajectories, results -- compared across builds
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import conftest
api = conftest.load_api()
out = sys.argv[1]
res = {}
p = api.manipulator7_problem(api.SOLVER_IPDDP, 150, True, 16)
B = 1024
x0 = api.batch_x0(p, B, 1234, 0.02 * np.ones(p.nx))
hs = api.HipBatchSolver(p, B); hs.set_initial(x0); hs.initialize(); hs.backward()
K, k = hs.gains(); Vx, Vxx = hs.value()
res["K1"] = K; res["k1"] = k; res["Vx1"] = Vx; res["Vxx1"] = Vxx
hs.close()
hs = api.HipBatchSolver(p, B); hs.set_initial(x0)
st = hs.solve()
K, k = hs.gains(); Vx, Vxx = hs.value(); X, U = hs.trajectory(); r = hs.results()
res["K"] = K; res["k"] = k; res["Vx"] = Vx; res["X"] = X; res["U"] = U; res["r"] = r.view(np.uint8)
print("iterations", st.outer_iterations if hasattr(st, "outer_iterations") else st)
hs.close()
np.savez(out, **res)
print("saved", out)
