#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/*.json from the CPU oracle (oracle/, the Eigen-free restatement of
the reference path; SURVEY.md section 8(c): the reference itself cannot be built here, so these vectors pin the
restatement, not the Eigen binary -- "parity unpinned" stays true).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.json

Each fixture holds, for the example trajectory of a config (trajectory 0 = the reference example's x0):
  * the problem name (built by tests/test_gpu_parity.make), x0;
  * step level: K, k, V_x, V_xx at t in {0, N/2, N-1} and dV, reg of the FIRST backward pass;
  * solve level: the per-iteration history (objective, merit, alpha_pr, alpha_du, inf_du, inf_pr, inf_comp, mu,
    regularization), the result record, the final controls at t in {0, N/2, N-1}.
Floats are written with repr() (exact round trip)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from conftest import load_api          # noqa: E402
import test_gpu_parity as T            # noqa: E402

CASES = ["pendulum_ipddp_unc", "pendulum_ipddp_box", "pendulum_clddp_unc", "pendulum_clddp_box",
         "cartpole_ipddp_box", "cartpole_clddp_box", "unicycle_ipddp_box_ball",
         "term_ineq_only", "term_eq_only", "path_term_eq", "path_term_ineq"]


def tolist(a):
    return np.asarray(a, dtype=np.float64).tolist()


def main():
    api = load_api()
    for case in CASES:
        p = T.make(api, case)
        p.options.return_iteration_info = 1
        x0 = np.array(p.x0, dtype=np.float64)
        U0 = api.batch_U0(p, 1)
        N = p.N
        ts = sorted({0, N // 2, N - 1})
        o = api.Oracle(p); o.set_initial(x0, None if U0 is None else U0[0]); o.initialize()
        ok = o.backward()
        K, k = o.gains(); Vx, Vxx = o.value(); dV, reg = o.backward_scalars()
        step = {"ok": int(bool(ok)), "t": ts, "K": tolist(K[ts]), "k": tolist(k[ts]), "Vx": tolist(Vx[ts]), "Vxx": tolist(Vxx[ts]),
                "dV": tolist(dV), "reg": float(reg)}
        o2 = api.Oracle(p); o2.set_initial(x0, None if U0 is None else U0[0])
        r = o2.solve()
        X, U = o2.trajectory()
        sol = {"history": tolist(o2.history()), "iterations": int(r["iterations"]), "status": int(r["status"]),
               "final_objective": float(r["final_objective"]), "n_backward": int(r["n_backward"]), "n_forward": int(r["n_forward"]),
               "U": tolist(U[ts]), "X_final": tolist(X[N])}
        out = {"case": case, "x0": tolist(x0), "horizon": int(N), "step": step, "solve": sol,
               "generator": "tests/golden/make_golden.py (oracle/cddp_oracle.cpp, -O2 -ffp-contract=off)"}
        fn = os.path.join(HERE, case + ".json")
        json.dump(out, open(fn, "w"), indent=None, separators=(",", ":"))
        print(case, "iterations", sol["iterations"], "status", sol["status"], "bytes", os.path.getsize(fn))


if __name__ == "__main__":
    main()
