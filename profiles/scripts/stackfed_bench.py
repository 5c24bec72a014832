"""Kernel time and streaming rate of the stack-fed sweeps (cddp_hip_stacks_backward) at the C2 shape: nx 4, nu 1, N 100, batch 4096.
Random well-conditioned stacks (the arithmetic does not depend on the values); bytes = the stacks read + the gains / value stacks
written, the sect. 8(d) model of the backward class.  Run on the GPU box:  python profiles/scripts/stackfed_bench.py out.md"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import numpy as np
from conftest import load_api

api = load_api()
rng = np.random.default_rng(1)
rows = []
for (nx, nu, m, N, B) in ((4, 1, 2, 100, 4096), (3, 2, 5, 200, 8192), (12, 4, 8, 400, 2048), (14, 7, 14, 150, 4096)):
    fx = np.tile(np.eye(nx), (B, N, 1, 1)) + 0.05 * rng.standard_normal((B, N, nx, nx)); fu = 0.1 * rng.standard_normal((B, N, nx, nu))
    lx = rng.standard_normal((B, N, nx)); lu = rng.standard_normal((B, N, nu))
    lxx = np.tile(np.eye(nx), (B, N, 1, 1)); luu = np.tile(np.eye(nu), (B, N, 1, 1)); lux = np.zeros((B, N, nu, nx))
    VxN = rng.standard_normal((B, nx)); VxxN = np.tile(10.0 * np.eye(nx), (B, 1, 1))
    opt = api.default_options()
    for branch, mm, name in ((api.STACKS_IPDDP, 0, "IPDDP, no constraints"), (api.STACKS_LOGDDP, 0, "LogDDP"), (api.STACKS_CLDDP, 0, "CLDDP, no bounds"),
                             (api.STACKS_IPDDP_PATH, m, "IPDDP, path constraints")):
        try:
            hs = api.HipStackSolver(B, nx, nu, mm, N)
        except api.HipError:
            continue
        hs.set_stacks(fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN)
        mu = None
        if mm:
            y = np.full((B, N, mm), 0.5); s = np.full((B, N, mm), 0.4); g = -s + 0.01 * rng.standard_normal((B, N, mm))
            Gx = 0.1 * rng.standard_normal((B, N, mm, nx)); Gu = 0.3 * rng.standard_normal((B, N, mm, nu))
            hs.set_constraint_stacks(y, s, g, Gx, Gu); mu = np.full(B, 0.1)
        reg = np.full(B, 1e-6)
        dyn = nx * nx + nx * nu; cost = nx + nu + nx * nx + nu * nu + nu * nx; gain = nu * nx + nu; val = nx + nx * nx
        con = (3 * mm + mm * nx + mm * nu) + (2 * mm + 2 * mm * nx) if mm else 0
        bytes_ = 8.0 * B * (N * (dyn + cost + gain + val + con) + val)
        for form in ("lane", "coop"):   # one lane per trajectory | sixteen lanes per trajectory (stacks_coop.hpp)
            os.environ["CDDP_HIP_STACKS_SWEEP"] = form
            ms = []
            for _ in range(4):
                ok = hs.backward(branch, opt, reg, mu, retry=False); ms.append(hs.kernel_ms())
            if hs.sweep_form() != (1 if form == "coop" else 0):
                continue   # this form is not instantiated for the shape
            t = min(ms[1:])
            rows.append((nx, nu, mm, N, B, name, form, t, bytes_ / t / 1e6, int(ok.sum())))
            print(rows[-1], flush=True)
        os.environ.pop("CDDP_HIP_STACKS_SWEEP", None)
        hs.close()
with open(sys.argv[1], "w") as f:
    f.write("# Stack-fed sweeps (host plug-in mode): kernel time per sweep of a whole batch, one-lane and lane-cooperative forms, MI355X, round 3\n\n")
    f.write("`python profiles/scripts/stackfed_bench.py`; hipEvent time of the single launch (`cddp_hip_stacks_last_kernel_ms`), best of 3; GB/s = the "
            "backward class's algorithmic bytes of DESIGN.md section 4 over that time.\n\n| nx | nu | m | N | batch | branch | form | ms | GB/s | sweeps ok |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows: f.write("| %d | %d | %d | %d | %d | %s | %s | %.3f | %.0f | %d |\n" % r)
