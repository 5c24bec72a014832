"""The C++ oracle (and, on the GPU box, the HIP path) against the numpy twin's committed vectors.

tests/golden/twin_<case>.json come from oracle/twin/cddp_twin.py, a second restatement of the reference path written
independently of oracle/cddp_oracle.cpp (tests/golden/make_twin_golden.py).  Two independent readings of the reference
must agree: step level (one sweep from the initial guess: K, k, V_x, V_xx, dV, every trial of the ladder) at 1e-11
relative, solve level in iteration count, status, sweep / rollout counts and the per-iteration history
(objective, merit, alpha_pr, alpha_du, inf_du, inf_pr, inf_comp, mu, regularisation).

This does not lift "parity unpinned" (no Eigen / autodiff build of the reference exists here); it removes the
single-reading risk -- see DESIGN.md section 5.
"""
import glob
import json
import os

import numpy as np
import pytest

from test_gpu_parity import TERM_CASES, make, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "twin_*.json")))
NAMES = [os.path.basename(f)[len("twin_"):-len(".json")] for f in FIXTURES]

STEP_TOL = 1e-11     # two float64 implementations with different summation orders (numpy BLAS vs scalar loops)
HIST_TOL = 1e-7      # history entries pass through up to 100 nonlinear iterations


def _load(name):
    with open(os.path.join(HERE, "golden", "twin_%s.json" % name)) as f:
        return json.load(f)


def _problem(api, name):
    return TERM_CASES[name](api) if name in TERM_CASES else make(api, name)


def _num(v):
    return np.inf if v is None else v


def test_fixtures_present():
    assert len(NAMES) >= 14, NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_twin_step_level(api, oracle_built, name):
    fx = _load(name)
    p = _problem(api, name)
    o = api.Oracle(p)
    assert list(o.alphas()) == fx["alphas"]
    U0 = api.batch_U0(p, 1)
    o.set_initial(p.x0, None if U0 is None else U0[0])
    o.initialize()
    assert rel_err(o.result()["final_objective"], fx["init"]["cost"]) < STEP_TOL
    ok = o.backward(retry=True)
    sw = fx["sweep"]
    assert bool(ok) == sw["ok"]
    K, k = o.gains(); Vx, Vxx = o.value(); dV, reg = o.backward_scalars()
    assert reg == sw["reg"]
    for i, t in enumerate(sw["t"]):
        assert rel_err(K[t], sw["K"][i]) < STEP_TOL, (name, t)
        assert rel_err(k[t], sw["k"][i]) < STEP_TOL, (name, t)
        assert rel_err(Vx[t], sw["Vx"][i]) < STEP_TOL, (name, t)
        assert rel_err(Vxx[t], sw["Vxx"][i]) < STEP_TOL, (name, t)
    assert rel_err(np.sum(K), sw["K_sum"]) < 1e-9 and rel_err(np.sum(Vxx), sw["Vxx_sum"]) < 1e-9   # every step, not only the sampled ones
    assert rel_err(dV, sw["dV"]) < STEP_TOL
    r = o.result()
    assert rel_err(r["inf_du"], sw["inf_du"]) < STEP_TOL and rel_err(r["step_norm"], sw["step_norm"]) < STEP_TOL
    if sw["inf_pr"] is not None:
        assert rel_err(r["inf_pr"], sw["inf_pr"]) < STEP_TOL and rel_err(r["inf_comp"], sw["inf_comp"]) < STEP_TOL
    for tr in fx["trials"]:
        t = o.forward(tr["alpha"])
        assert bool(t["success"]) == tr["success"], (name, tr["alpha"], t, tr)
        assert abs(t["alpha_pr"] - tr["alpha_pr"]) < 1e-12
        if p.c.solver == api.SOLVER_IPDDP:
            assert abs(t["alpha_du"] - tr["alpha_du"]) < 1e-12
        if tr["success"]:
            assert rel_err(t["cost"], tr["cost"]) < STEP_TOL and rel_err(t["merit_function"], tr["merit"]) < STEP_TOL
            if tr["theta"] is not None and p.c.solver == api.SOLVER_IPDDP:
                assert rel_err(t["theta"], tr["theta"]) < STEP_TOL


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_twin_solve_level(api, oracle_built, name):
    fx = _load(name)["solve"]
    p = _problem(api, name)
    p.options.return_iteration_info = 1
    o = api.Oracle(p)
    U0 = api.batch_U0(p, 1)
    o.set_initial(p.x0, None if U0 is None else U0[0])
    r = o.solve()
    assert (r["iterations"], r["status"], r["n_backward"], r["n_forward"]) == (fx["iterations"], fx["status"], fx["n_backward"], fx["n_forward"]), (name, r)
    assert rel_err(r["final_objective"], fx["final_objective"]) < HIST_TOL
    h = o.history(); hf = np.array(fx["history"])
    assert h.shape == hf.shape, (h.shape, hf.shape)
    h = np.where(np.isfinite(h), h, -1.0)
    if p.c.solver == api.SOLVER_CLDDP:
        h[:, 7] = 0.0; hf[:, 7] = 0.0     # barrier_mu is not recorded by CLDDP (cddp_solver_base.cpp:228-230)
    assert rel_err(h, hf) < HIST_TOL, (name, np.max(np.abs(h - hf) / np.maximum(1.0, np.abs(hf)), axis=0))
    X, U = o.trajectory(); K, _ = o.gains()
    assert rel_err(U[0], fx["U_first"]) < 1e-6 and rel_err(U[-1], fx["U_last"]) < 1e-6 and rel_err(X[-1], fx["xN"]) < 1e-6
    assert rel_err(K[0], fx["K0"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_twin(api, name):
    """The HIP path against the twin's vectors directly (no C++ oracle in between): sweep 1e-8, solve-level counts."""
    fx = _load(name)
    p = _problem(api, name)
    U0 = api.batch_U0(p, 1)
    hs = api.HipBatchSolver(p, 1)
    hs.set_initial(p.x0[None, :], U0)
    hs.initialize()
    ok = hs.backward()
    sw = fx["sweep"]
    assert bool(ok[0]) == sw["ok"]
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    assert reg[0] == sw["reg"]
    for i, t in enumerate(sw["t"]):
        assert max(rel_err(K[0, t], sw["K"][i]), rel_err(k[0, t], sw["k"][i]), rel_err(Vx[0, t], sw["Vx"][i]), rel_err(Vxx[0, t], sw["Vxx"][i])) < 1e-8
    assert rel_err(dV[0], sw["dV"]) < 1e-8
    trials = hs.forward(np.array(fx["alphas"]))
    for a, tr in enumerate(fx["trials"]):
        g = trials[0, a]
        assert bool(g["success"]) == tr["success"], (name, tr["alpha"])
        if tr["success"]:
            assert rel_err(g["cost"], tr["cost"]) < 1e-8 and rel_err(g["merit_function"], tr["merit"]) < 1e-8
    hs.close()
    p2 = _problem(api, name)
    hs = api.HipBatchSolver(p2, 1)
    hs.set_initial(p2.x0[None, :], U0)
    hs.solve()
    r = hs.results()[0]
    fs = fx["solve"]
    assert (int(r["iterations"]), int(r["status"]), int(r["n_backward"]), int(r["n_forward"])) == (fs["iterations"], fs["status"], fs["n_backward"], fs["n_forward"]), (name, r)
    assert rel_err(r["final_objective"], fs["final_objective"]) < 1e-6
    hs.close()
