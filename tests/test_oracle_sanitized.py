"""The CPU checker under AddressSanitizer + UndefinedBehaviorSanitizer (VERDICT r03 "missing" #6): every solver of
oracle/cddp_oracle.cpp -- CLDDP, IPDDP (path constraints, terminal equality), LogDDP, MSIPDDP -- solves a small problem in a child
process that preloads libasan and loads `make -C oracle sanitize`'s library through CDDP_ORACLE_LIB.  A heap overflow, use after
free, signed overflow, misaligned or out-of-bounds index in the restatement aborts the child (-fno-sanitize-recover)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import conftest
api = conftest.load_api()
assert api.ORACLE_LIB_PATH.endswith("libcddp_oracle_san.so"), api.ORACLE_LIB_PATH
def run(p, B=3, seed=7):
    x0 = api.batch_x0(p, B, seed, 0.05 * np.ones(p.nx))
    res = api.oracle_solve_batch(p, x0, None, None, n_threads=2, want_traj=False)[0]
    assert np.all(res["iterations"] > 0), res
    return res
for solver in (api.SOLVER_CLDDP, api.SOLVER_IPDDP, api.SOLVER_LOGDDP):
    for mk in (lambda s: api.pendulum_problem(s, True, 40), lambda s: api.cartpole_problem(s, True, 30), lambda s: api.unicycle_problem(s, 40, True)):
        p = mk(solver); p.options.max_iterations = 12
        run(p)
p = api.cartpole_problem(api.SOLVER_IPDDP, True, 30); p.options.max_iterations = 8; p.options.use_ilqr = 0; run(p)
p = api.pendulum_problem(api.SOLVER_IPDDP, True, 30); p.add_terminal_equality("TerminalEq", [0.0, 0.0]); p.options.max_iterations = 8; run(p)
p = api.pendulum_problem(api.SOLVER_MSIPDDP, True, 40); p.options.max_iterations = 8; run(p)
p = api.cartpole_problem(api.SOLVER_MSIPDDP, False, 30); p.options.max_iterations = 8; run(p)
for trig in (1, 0):
    api.set_trig_mode(trig)
    p = api.cartpole_problem(api.SOLVER_LOGDDP, True, 30); p.options.max_iterations = 6; p.options.enable_parallel = 1; run(p)
print("SANITIZED-OK")
'''


def test_oracle_solvers_under_asan_ubsan():
    lib = os.path.join(REPO, "oracle", "_build", "libcddp_oracle_san.so")
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan, CDDP_ORACLE_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
