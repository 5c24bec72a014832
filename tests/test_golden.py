"""Golden fixtures (tests/golden/*.json, written by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle reproduces its own committed vectors (guards the restatement against silent drift).
GPU: the HIP path reproduces them through the C-ABI -- gains / value within 1e-8, identical iteration count,
status and sweep / rollout counts, per-iteration history within 1e-8."""
import glob
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
# (twin_*.json are the numpy twin's vectors -> tests/test_twin_golden.py; trig_noise_flip_rates.json -> test_oracle_trig_noise.py;
#  portfolio_thresholds.json -> test_pycddp_portfolio.py; ref_boxqp_inputs.json -> test_reference_plant_pins.py)
FIXTURES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.json"))
                  if not os.path.basename(f).startswith(("twin_", "trig_noise", "portfolio_", "ref_")))
TOL_GPU = 1e-8
TOL_CPU = 1e-11   # same code, possibly another libm build


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    a = np.where(same_inf, 0.0, a); b = np.where(same_inf, 0.0, b)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def load(fn):
    return json.load(open(fn))


def problem(api, fx):
    import test_gpu_parity as T
    p = T.make(api, fx["case"])
    p.options.return_iteration_info = 1
    return p


def test_fixtures_present():
    assert len(FIXTURES) >= 10


@pytest.mark.parametrize("fn", FIXTURES, ids=[os.path.basename(f)[:-5] for f in FIXTURES])
def test_oracle_reproduces_golden(api, oracle_built, fn):
    fx = load(fn)
    p = problem(api, fx)
    x0 = np.array(fx["x0"]); U0 = api.batch_U0(p, 1)
    ts = fx["step"]["t"]
    o = api.Oracle(p); o.set_initial(x0, None if U0 is None else U0[0]); o.initialize()
    assert int(bool(o.backward())) == fx["step"]["ok"]
    K, k = o.gains(); Vx, Vxx = o.value(); dV, reg = o.backward_scalars()
    for name, got in (("K", K[ts]), ("k", k[ts]), ("Vx", Vx[ts]), ("Vxx", Vxx[ts]), ("dV", dV)):
        assert rel(got, fx["step"][name]) < TOL_CPU, name
    assert reg == fx["step"]["reg"]
    o2 = api.Oracle(p); o2.set_initial(x0, None if U0 is None else U0[0])
    r = o2.solve()
    s = fx["solve"]
    assert int(r["iterations"]) == s["iterations"] and int(r["status"]) == s["status"]
    assert int(r["n_backward"]) == s["n_backward"] and int(r["n_forward"]) == s["n_forward"]
    assert rel(r["final_objective"], s["final_objective"]) < TOL_CPU
    h = o2.history()
    assert h.shape == np.asarray(s["history"]).reshape(-1, 9).shape
    assert rel(h, np.asarray(s["history"]).reshape(-1, 9)) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("fn", FIXTURES, ids=[os.path.basename(f)[:-5] for f in FIXTURES])
def test_hip_reproduces_golden(api, fn):
    fx = load(fn)
    p = problem(api, fx)
    x0 = np.array(fx["x0"])[None, :]; U0 = api.batch_U0(p, 1)
    ts = fx["step"]["t"]
    hs = api.HipBatchSolver(p, 1); hs.set_initial(x0, U0); hs.initialize()
    ok = hs.backward()
    assert int(ok[0]) == fx["step"]["ok"]
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    for name, got in (("K", K[0][ts]), ("k", k[0][ts]), ("Vx", Vx[0][ts]), ("Vxx", Vxx[0][ts]), ("dV", dV[0])):
        assert rel(got, fx["step"][name]) < TOL_GPU, name
    assert reg[0] == fx["step"]["reg"]
    hs.close()
    hs = api.HipBatchSolver(p, 1); hs.set_initial(x0, U0); hs.solve()
    r = hs.results()[0]
    s = fx["solve"]
    assert int(r["iterations"]) == s["iterations"] and int(r["status"]) == s["status"]
    assert int(r["n_backward"]) == s["n_backward"] and int(r["n_forward"]) == s["n_forward"]
    assert rel(r["final_objective"], s["final_objective"]) < 1e-7
    h = hs.history(1)[0]
    g = np.asarray(s["history"]).reshape(-1, 9)
    assert h.shape == g.shape
    if s["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE):   # converged solves: the whole trace
        assert rel(h, g) < 1e-6
    X, U = hs.trajectory()
    assert rel(U[0][ts], s["U"]) < 1e-5
    hs.close()
