"""Lane-cooperative stack-fed sweep (cddp-cpp_amd/csrc/stacks_coop.hpp, the default for nx > 8) against the one-lane form
(stacks.hip::sweep) on the same uploaded stacks: every output BITWISE equal, for every branch the boundary offers
(clddp_solver.cpp:79-204 with and without the control box, ipddp_solver.cpp:1048-1118 and 1355-1568, logddp_solver.cpp:470-575,
msipddp_solver.cpp:1112-1208), with and without the dynamics Hessian stacks, and through the "increase the regularisation and
retry" loop (cddp_solver_base.cpp:93-111).  The one-lane form itself is held to the oracle / twins by tests/test_stack_fed*.py and
tests/test_logddp_stack_fed.py; (12, 4, 8) additionally replays one quadrotor-shaped sweep of the solver core."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(4, 1, 0), (4, 1, 2), (3, 2, 0), (3, 2, 5), (6, 3, 0), (6, 3, 6), (12, 4, 0), (12, 4, 8), (13, 4, 8)]


def make_stacks(rng, B, N, nx, nu, bad=()):
    fx = np.tile(np.eye(nx), (B, N, 1, 1)) + 0.05 * rng.standard_normal((B, N, nx, nx))
    fu = 0.1 * rng.standard_normal((B, N, nx, nu))
    lx = rng.standard_normal((B, N, nx)); lu = rng.standard_normal((B, N, nu))
    W = 0.2 * rng.standard_normal((B, N, nx, nx)); lxx = np.tile(np.eye(nx), (B, N, 1, 1)) + W @ np.swapaxes(W, 2, 3)
    W = 0.2 * rng.standard_normal((B, N, nu, nu)); luu = np.tile(np.eye(nu), (B, N, 1, 1)) + W @ np.swapaxes(W, 2, 3)
    lux = 0.05 * rng.standard_normal((B, N, nu, nx))
    for b in bad:   # an indefinite Q_uu late in the horizon: the first sweep of these trajectories fails
        luu[b, N - 3] = -5.0 * np.eye(nu)
    VxN = rng.standard_normal((B, nx))
    W = rng.standard_normal((B, nx, nx)); VxxN = 4.0 * np.tile(np.eye(nx), (B, 1, 1)) + 0.3 * (W + 0.5 * np.swapaxes(W, 1, 2))   # NOT symmetric
    return fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN


def run_both(hs, api, branch, opt, reg, mu, retry, path):
    out = {}
    for form in ("lane", "coop"):
        os.environ["CDDP_HIP_STACKS_SWEEP"] = form
        try:
            ok = hs.backward(branch, opt, reg, mu, retry=retry)
        finally:
            os.environ.pop("CDDP_HIP_STACKS_SWEEP", None)
        assert hs.sweep_form() == (1 if form == "coop" else 0)
        res = {"ok": ok.copy()}
        for name, v in zip(("K", "k", "Vx", "Vxx", "dV"), hs.gains()): res[name] = v
        res.update(hs.scalars())
        if path:
            for name, v in zip(("ky", "Ky", "ks", "Ks", "dX"), hs.constraint_gains()): res[name] = v
        out[form] = res
    return out["lane"], out["coop"]


def assert_bitwise(lane, coop, only_ok=True):
    good = lane["ok"].astype(bool)
    assert np.array_equal(lane["ok"], coop["ok"])
    for name in lane:
        a, c = lane[name], coop[name]
        if name in ("ok",):
            continue
        if only_ok and name not in ("reg",):   # a failed sweep leaves partial stacks behind; the forms stop at the same step but
            a, c = a[good], c[good]            # the value stacks of the failing step differ by which lanes had stored already
        assert np.array_equal(a, c, equal_nan=True), "%s differs between the one-lane and the cooperative sweep (max |d| = %g)" % (
            name, float(np.nanmax(np.abs(a - c))))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "nx%d_nu%d_m%d" % s)
def test_cooperative_sweep_is_bitwise_the_lane_sweep(api, shape):
    nx, nu, m = shape
    rng = np.random.default_rng(100 * nx + 10 * nu + m)
    B, N = (9, 12) if nx >= 12 else (70, 25)   # 70: more than one wavefront of groups, a ragged tail
    opt = api.default_options()
    try:
        hs = api.HipStackSolver(B, nx, nu, m, N)
    except api.HipError as e:
        pytest.skip(str(e))
    stacks = make_stacks(rng, B, N, nx, nu)
    hs.set_stacks(*stacks)
    reg = np.where(np.arange(B) % 2 == 0, 1e-6, 1e-2)
    if m:
        y = 0.5 + 0.4 * rng.random((B, N, m)); s = 0.2 + 0.4 * rng.random((B, N, m)); g = -s + 0.01 * rng.standard_normal((B, N, m))
        Gx = 0.1 * rng.standard_normal((B, N, m, nx)); Gu = 0.3 * rng.standard_normal((B, N, m, nu))
        hs.set_constraint_stacks(y, s, g, Gx, Gu)
        mu = np.where(np.arange(B) % 3 == 0, 0.1, 1e-3)
        lane, coop = run_both(hs, api, api.STACKS_IPDDP_PATH, opt, reg, mu, False, True)
        assert lane["ok"].all()
        assert_bitwise(lane, coop)
        assert np.all(coop["alpha_pr_max"] > 0) and np.any(coop["alpha_pr_max"] < 1.0)
    else:
        for branch in (api.STACKS_IPDDP, api.STACKS_LOGDDP, api.STACKS_CLDDP):
            lane, coop = run_both(hs, api, branch, opt, reg, None, False, False)
            assert lane["ok"].all(), branch
            assert_bitwise(lane, coop)
        # multiple-shooting defects
        hs.set_defect_stack(0.05 * rng.standard_normal((B, N, nx)))
        lane, coop = run_both(hs, api, api.STACKS_MSIPDDP, opt, reg, None, False, False)
        assert lane["ok"].all()
        assert_bitwise(lane, coop)
        hs.set_defect_stack(None)
        # CLDDP control box: clamped and free directions, warm start from the previous sweep's k
        U = 0.98 * np.sign(rng.standard_normal((B, N, nu)))   # next to a bound: steps towards it are clamped
        hs.set_control_box(-np.ones(nu), np.ones(nu), U)
        for _ in range(2):
            lane, coop = run_both(hs, api, api.STACKS_CLDDP, opt, reg, None, False, False)
            assert_bitwise(lane, coop)
        assert np.any(np.all(coop["K"] == 0.0, axis=3)) and np.any(coop["K"] != 0.0)   # some rows clamped, some free
        hs.set_control_box(None, None, None)
    hs.close()


@pytest.mark.parametrize("shape", [(4, 1, 2), (3, 2, 0), (6, 3, 6), (12, 4, 0), (12, 4, 8), (13, 4, 8), (14, 7, 0)], ids=lambda s: "nx%d_nu%d_m%d" % s)
def test_large_state_sweeps_on_the_batch_minor_layout(api, shape, monkeypatch):
    """Round 6: handles whose default sweep is the cooperative one (nx > 8) keep their stacks tile-minor, [t][b / 4][e][b % 4] (the step record of
    the four trajectories of a workgroup contiguous); CDDP_HIP_STACKS_LAYOUT=plain keeps [t][e][batch] -- both kernels read both layouts, and
    every comparison above ran on the default.  Same comparison on the other layout, and the two layouts against each other."""
    nx, nu, m = shape
    for layout in ("plain", "t4"):   # (the default of a handle depends on its shape and batch: both, explicitly)
        monkeypatch.setenv("CDDP_HIP_STACKS_LAYOUT", layout)
        if shape in SHAPES:   # (its closing assertion on clamped / free BoxQP rows is tuned to these shapes' data)
            test_cooperative_sweep_is_bitwise_the_lane_sweep(api, shape)
    rng = np.random.default_rng(7)
    B, N = 11, 9
    stacks = make_stacks(rng, B, N, nx, nu)
    opt = api.default_options(); reg = np.full(B, 1e-6)
    out = {}
    for layout in ("plain", "t4"):
        monkeypatch.setenv("CDDP_HIP_STACKS_LAYOUT", layout)
        hs = api.HipStackSolver(B, nx, nu, m, N)
        hs.set_stacks(*stacks)
        mu = None
        if m:
            r2 = np.random.default_rng(8)
            y = 0.5 + 0.4 * r2.random((B, N, m)); s = 0.2 + 0.4 * r2.random((B, N, m)); g = -s + 0.01 * r2.standard_normal((B, N, m))
            hs.set_constraint_stacks(y, s, g, 0.1 * r2.standard_normal((B, N, m, nx)), 0.3 * r2.standard_normal((B, N, m, nu)))
            mu = np.full(B, 0.05)
        monkeypatch.setenv("CDDP_HIP_STACKS_SWEEP", "coop")
        ok = hs.backward(api.STACKS_IPDDP_PATH if m else api.STACKS_IPDDP, opt, reg, mu, retry=False)
        assert ok.all() and hs.sweep_form() == 1
        out[layout] = list(hs.gains()) + (list(hs.constraint_gains()) if m else [])
        hs.close()
    for a, c in zip(out["plain"], out["t4"]):
        assert np.array_equal(a, c)


@pytest.mark.parametrize("shape", [(4, 1, 0), (3, 2, 5), (6, 3, 6), (12, 4, 0), (12, 4, 8)], ids=lambda s: "nx%d_nu%d_m%d" % s)
def test_cooperative_sweep_hessian_stacks(api, shape):
    """Full DDP (use_ilqr = false): the dt-scaled dynamics Hessian tensors weighted with V_x (ipddp_solver.cpp:1070-1082, 1396-1408;
    logddp_solver.cpp:505-515)."""
    nx, nu, m = shape
    rng = np.random.default_rng(7 + nx)
    B, N = (6, 10) if nx >= 12 else (40, 16)
    opt = api.default_options()
    hs = api.HipStackSolver(B, nx, nu, m, N)
    hs.set_stacks(*make_stacks(rng, B, N, nx, nu))
    reg = np.full(B, 1e-6); mu = None
    branches = (api.STACKS_IPDDP, api.STACKS_LOGDDP)
    if m:
        y = 0.5 + 0.4 * rng.random((B, N, m)); s = 0.2 + 0.4 * rng.random((B, N, m)); g = -s + 0.01 * rng.standard_normal((B, N, m))
        hs.set_constraint_stacks(y, s, g, 0.1 * rng.standard_normal((B, N, m, nx)), 0.05 * rng.standard_normal((B, N, m, nu)))
        mu = np.full(B, 1e-2); branches = (api.STACKS_IPDDP_PATH,)
    plain = {}
    for branch in branches:
        plain[branch] = run_both(hs, api, branch, opt, reg, mu, False, bool(m))[1]
    hs.set_hessian_stacks(0.02 * rng.standard_normal((B, N, nx, nx, nx)), 0.02 * rng.standard_normal((B, N, nx, nu, nu)),
                          0.02 * rng.standard_normal((B, N, nx, nu, nx)))
    for branch in branches:
        lane, coop = run_both(hs, api, branch, opt, reg, mu, False, bool(m))
        assert lane["ok"].all()
        assert_bitwise(lane, coop)
        assert not np.array_equal(coop["K"], plain[branch]["K"])   # the tensors did enter
    hs.close()


@pytest.mark.parametrize("shape", [(4, 1, 0), (3, 2, 0), (6, 3, 0), (12, 4, 0)], ids=lambda s: "nx%d_nu%d_m%d" % s)
def test_cooperative_sweep_retry_loop(api, shape):
    """The reference's backward passes fail on an indefinite Q_uu only in CLDDP (eigenvalue test, clddp_solver.cpp:133-139; Eigen's LDLT
    reports Success for indefinite blocks): trajectories with an indefinite step fail a single attempt and come back from the retry loop
    (cddp_solver_base.cpp:93-111) with a larger regularisation -- the same one, and the same gains, in both forms."""
    nx, nu, m = shape
    rng = np.random.default_rng(17 + nx)
    B, N = (6, 10) if nx >= 12 else (40, 16)
    opt = api.default_options()
    hs = api.HipStackSolver(B, nx, nu, m, N)
    bad = (1, 4)
    hs.set_stacks(*make_stacks(rng, B, N, nx, nu, bad=bad))
    reg = np.full(B, 1e-6)
    lane, coop = run_both(hs, api, api.STACKS_CLDDP, opt, reg, None, False, False)
    assert not lane["ok"][list(bad)].any() and lane["ok"].sum() == B - len(bad)
    assert_bitwise(lane, coop)
    lane, coop = run_both(hs, api, api.STACKS_CLDDP, opt, reg, None, True, False)
    assert lane["ok"].all()
    assert np.all(lane["reg"][list(bad)] > 1.0) and np.all(np.delete(lane["reg"], bad) == 1e-6)
    assert_bitwise(lane, coop)
    hs.close()


def test_large_state_defaults_to_the_cooperative_form(api):
    rng = np.random.default_rng(3)
    opt = api.default_options()
    for (nx, nu, m), want in (((12, 4, 0), 1), ((4, 1, 0), 0), ((6, 3, 0), 1)):
        hs = api.HipStackSolver(5, nx, nu, m, 6)
        hs.set_stacks(*make_stacks(rng, 5, 6, nx, nu))
        os.environ.pop("CDDP_HIP_STACKS_SWEEP", None)
        assert hs.backward(api.STACKS_IPDDP, opt, np.full(5, 1e-6)).all()
        assert hs.sweep_form() == want
        hs.close()
    # round 6: small shapes with path rows take the cooperative form while the batch leaves the chip mostly empty under the one-lane one
    for B, want in ((70, 1), (8192 + 64, 0)):
        hs = api.HipStackSolver(B, 4, 1, 2, 3)
        hs.set_stacks(*make_stacks(rng, B, 3, 4, 1))
        hs.set_constraint_stacks(np.full((B, 3, 2), 0.5), np.full((B, 3, 2), 0.4), np.full((B, 3, 2), -0.39), np.zeros((B, 3, 2, 4)), np.tile(np.array([[1.0], [-1.0]]), (B, 3, 1, 1)))
        assert hs.backward(api.STACKS_IPDDP_PATH, opt, np.full(B, 1e-6), np.full(B, 0.1)).all()
        assert hs.sweep_form() == want
        hs.close()
    # a shape that exists only in the cooperative form (the 7-joint arm with its control box: the one-lane kernel would need
    # ~40 KB of scratch per lane)
    hs = api.HipStackSolver(5, 14, 7, 14, 6)
    hs.set_stacks(*make_stacks(rng, 5, 6, 14, 7))
    y = np.full((5, 6, 14), 0.5); s = np.full((5, 6, 14), 0.4)
    hs.set_constraint_stacks(y, s, -s, np.zeros((5, 6, 14, 14)), np.concatenate([np.eye(7), -np.eye(7)])[None, None].repeat(5, 0).repeat(6, 1))
    assert hs.backward(api.STACKS_IPDDP_PATH, opt, np.full(5, 1e-6), np.full(5, 0.1)).all()
    assert hs.sweep_form() == 1
    hs.close()
