"""One shape / branch / form of the stack-fed sweep, a few launches -- the target of the rocprofv3 counter passes of
profiles/scripts/stackfed_pmc.sh.  usage: stackfed_one.py nx nu m N B form"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import numpy as np
from conftest import load_api

nx, nu, m, N, B = (int(v) for v in sys.argv[1:6]); form = sys.argv[6]
api = load_api(); rng = np.random.default_rng(1)
fx = np.tile(np.eye(nx), (B, N, 1, 1)) + 0.05 * rng.standard_normal((B, N, nx, nx)); fu = 0.1 * rng.standard_normal((B, N, nx, nu))
lx = rng.standard_normal((B, N, nx)); lu = rng.standard_normal((B, N, nu))
lxx = np.tile(np.eye(nx), (B, N, 1, 1)); luu = np.tile(np.eye(nu), (B, N, 1, 1)); lux = np.zeros((B, N, nu, nx))
VxN = rng.standard_normal((B, nx)); VxxN = np.tile(10.0 * np.eye(nx), (B, 1, 1))
opt = api.default_options()
hs = api.HipStackSolver(B, nx, nu, m, N)
hs.set_stacks(fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN)
mu = None; branch = api.STACKS_IPDDP
if m:
    y = np.full((B, N, m), 0.5); s = np.full((B, N, m), 0.4); g = -s + 0.01 * rng.standard_normal((B, N, m))
    hs.set_constraint_stacks(y, s, g, 0.1 * rng.standard_normal((B, N, m, nx)), 0.3 * rng.standard_normal((B, N, m, nu)))
    mu = np.full(B, 0.1); branch = api.STACKS_IPDDP_PATH
os.environ["CDDP_HIP_STACKS_SWEEP"] = form
for _ in range(3):
    ok = hs.backward(branch, opt, np.full(B, 1e-6), mu, retry=False)
    print(form, hs.sweep_form(), hs.kernel_ms(), int(ok.sum()), flush=True)
hs.close()
