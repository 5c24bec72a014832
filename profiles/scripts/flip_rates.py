"""Decision-flip statistics at a larger batch than the tests use (B = 256 per case): how many trajectories keep the oracle's
(status, iterations) on the HIP path, against the same count for the oracle under <= 1 ulp sin / cos noise and under <= 1 ulp
matrix-product noise (the two yardsticks of tests/test_oracle_trig_noise.py).  Run on the GPU box:
    python profiles/scripts/flip_rates.py gpurun_out/flip_rates_B256.md
The oracle is test infrastructure: this script is a measurement, not part of the product."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import numpy as np
from conftest import load_api
from test_gpu_parity import KNIFE_EDGE_CASES, TERM_CASES, make, spread_for

api = load_api()
lib = ctypes.CDLL(api.ORACLE_LIB_PATH)
B = 256
rows = []
for case in sorted(KNIFE_EDGE_CASES) + ["cartpole_ipddp_box", "unicycle_ipddp_box_ball", "unicycle_ipddp_box_state", "cartpole_clddp_box"]:
    p = TERM_CASES[case](api) if case in TERM_CASES else make(api, case)
    x0 = api.batch_x0(p, B, 20261201, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1)); U0 = api.batch_U0(p, B)
    X0 = np.tile(p.X0_single, (B, 1, 1)) if hasattr(p, "X0_single") else None
    if X0 is not None: X0[:, 0, :] = x0
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0, X0); hs.solve(); res = hs.results(); hs.close()
    ores = api.oracle_solve_batch(p, x0, U0, X0, n_threads=64, want_traj=False)[0]
    out = {}
    for name, setter in (("trig", lib.cddp_oracle_set_trig_noise), ("matmul", lib.cddp_oracle_set_matmul_noise)):
        try:
            setter(1)
            n = api.oracle_solve_batch(p, x0, U0, X0, n_threads=64, want_traj=False)[0]
        finally:
            setter(0)
        out[name] = int(((n["iterations"] == ores["iterations"]) & (n["status"] == ores["status"])).sum())
    same = (res["iterations"] == ores["iterations"]) & (res["status"] == ores["status"])
    conv = (ores["status"] == api.STATUS_OPTIMAL) | (ores["status"] == api.STATUS_ACCEPTABLE)
    both = same & conv
    oerr = max([abs(res["final_objective"][b] - ores["final_objective"][b]) / max(1.0, abs(ores["final_objective"][b])) for b in range(B) if both[b]] or [0.0])
    rows.append((case, int(same.sum()), out["trig"], out["matmul"], int(conv.sum()), oerr))
    print(rows[-1], flush=True)
with open(sys.argv[1], "w") as f:
    f.write("# Decision flips at B = 256 per case (HIP path vs oracle; oracle vs itself under <= 1 ulp noise), MI355X, round 2\n\n")
    f.write("`python profiles/scripts/flip_rates.py`: trajectories (of 256, seed 20261201) whose (status, iterations) equal the clean oracle's.\n\n")
    f.write("| case | HIP == oracle | oracle + trig noise == oracle | oracle + matmul noise == oracle | converged (oracle) | worst objective rel. err where both converge to the same counts |\n|---|---|---|---|---|---|\n")
    for r in rows: f.write("| `%s` | %d | %d | %d | %d | %.2e |\n" % r)
