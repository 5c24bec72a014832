#!/usr/bin/env python3
"""Reference pin, step 2 of 2 (test infrastructure): oracle/cddp_oracle.cpp against a cddp-cpp BINARY.

    oracle/ref_pin/build_ref.sh            # needs Eigen 3.4.0 + autodiff v1.1.2 checkouts (env vars), see its header
    python oracle/ref_pin/compare_traces.py [--write-fixtures]

Runs oracle/_ref/dump_traces (the reference's own solver core + plants, built from /root/reference where it lies) and the oracle
on BASELINE configs C1 - C3: the examples' own initial state (trajectory 0 of the bench batch) plus the next three seeded
initial states of bench.py's batch, IPDDP and CLDDP.  Compared per case:
    status string, iteration count                                   identical
    per-iteration history (objective, merit, alpha_pr / alpha_du, inf_du, inf_pr, inf_comp, mu, regularisation)   rel 1e-9
    final objective rel 1e-9, final trajectory rel 1e-7
    K_t, k_t (IPDDP also V_x, V_xx) at t in {0, N/2, N-1} of the last backward pass   rel 1e-8   (north_star: "gains within 1e-8")
With --write-fixtures the reference's outputs are stored under tests/golden/ref_trace_*.json (data, not source) so that the
`-m "not gpu"` suite holds the oracle to them from then on (tests/test_reference_traces.py picks them up when present).

This image has neither Eigen nor autodiff, so this script has never run green here: until someone runs it, every golden
fixture is the output of a restatement ("parity unpinned")."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(REPO, "oracle", "_ref", "dump_traces")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def rel(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    if a.shape != b.shape:
        return float("inf")
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


HIST_KEYS = ["objective", "merit_function", "step_length_primal", "step_length_dual", "dual_infeasibility", "primal_infeasibility",
             "complementary_infeasibility", "barrier_mu", "regularization"]   # = the oracle's history columns, oracle/cddp_oracle.cpp::recordHistory


def cases(api):
    out = []
    for name, mk, spread in (("pendulum", api.pendulum_problem, [0.1, 0.1]), ("cartpole", api.cartpole_problem, [0.1, 0.3, 0.1, 0.1]),
                             ("unicycle", lambda s, c=True: api.unicycle_problem(s, 200, True), [0.05, 0.05, 0.05])):
        for solver in ("IPDDP", "CLDDP"):
            if name == "unicycle" and solver == "CLDDP":
                continue   # CLDDP looks its box up under the name "ControlConstraint" (clddp_solver.cpp:147); the unicycle set-up names it "control_limits"
            p = mk(api.SOLVER_IPDDP if solver == "IPDDP" else api.SOLVER_CLDDP, True)
            p.options.return_iteration_info = 1
            x0s = api.batch_x0(p, 4, 20260928 + 1, spread)     # bench.py's seed: trajectories 0..3 of its batch
            out.append((name, solver, p, x0s))
    return out


def compare_one(api, name, solver, p, x0, ref):
    o = api.Oracle(p)
    U0 = api.batch_U0(p, 1)
    o.set_initial(x0, None if U0 is None else U0[0])
    r = o.solve()
    X, U = o.trajectory(); K, k = o.gains(); Vx, Vxx = o.value(); h = o.history()
    bad = []
    if api.STATUS_STRINGS[int(r["status"])] != ref["status"]:
        bad.append("status %s != %s" % (api.STATUS_STRINGS[int(r["status"])], ref["status"]))
    if int(r["iterations"]) != int(ref["iterations"]):
        bad.append("iterations %d != %d" % (int(r["iterations"]), int(ref["iterations"])))
    e = rel(r["final_objective"], ref["final_objective"])
    if e > 1e-9:
        bad.append("final objective rel %.2e" % e)
    for j, key in enumerate(HIST_KEYS):
        col = ref["history"][key]
        if len(col) != h.shape[0]:
            bad.append("history length %d != %d (%s)" % (h.shape[0], len(col), key)); break
        e = rel(h[:, j], col)
        if e > 1e-9:
            bad.append("history %s rel %.2e" % (key, e))
    if rel(X, ref["X"]) > 1e-7 or rel(U, ref["U"]) > 1e-7:
        bad.append("trajectory rel %.2e / %.2e" % (rel(X, ref["X"]), rel(U, ref["U"])))
    for ts, g in ref["gains"].items():
        t = int(ts)
        for nm, mine in (("K", K[t]), ("k", k[t])) + ((("V_x", Vx[t]), ("V_xx", Vxx[t])) if "V_xx" in g else ()):
            e = rel(mine, g[nm])
            if e > 1e-8:
                bad.append("%s[%d] rel %.2e" % (nm, t, e))
    return bad


def main():
    if not os.path.exists(BIN):
        raise SystemExit("%s is missing: run oracle/ref_pin/build_ref.sh first (needs Eigen 3.4.0 + autodiff v1.1.2; this image has neither)" % BIN)
    api = _load("cddp_cpp_amd_pyapi", os.path.join(REPO, "cddp-cpp_amd", "pyapi.py"))
    _load("cddp_oracle_api", os.path.join(REPO, "oracle", "oracle_api.py")).attach(api)
    api.set_trig_mode(0)     # glibc: the reference's own arithmetic
    write = "--write-fixtures" in sys.argv
    n_bad = 0
    for name, solver, p, x0s in cases(api):
        for b, x0 in enumerate(x0s):
            out = subprocess.run([BIN, name, solver] + ["%.17g" % v for v in x0], capture_output=True, text=True, check=True).stdout
            ref = json.loads(out.strip().splitlines()[-1])
            bad = compare_one(api, name, solver, p, x0, ref)
            print("%-9s %-5s traj %d  %-44s iterations %3d  %s" % (name, solver, b, ref["status"], ref["iterations"], "OK" if not bad else "MISMATCH: " + "; ".join(bad)))
            n_bad += bool(bad)
            if write:
                with open(os.path.join(REPO, "tests", "golden", "ref_trace_%s_%s_%d.json" % (name, solver.lower(), b)), "w") as f:
                    json.dump(ref, f)
    print("reference pin: %s" % ("GREEN -- the oracle reproduces the cddp-cpp binary on every case" if n_bad == 0 else "%d case(s) differ" % n_bad))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
