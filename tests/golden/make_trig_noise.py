"""Oracle-vs-noisy-oracle decision-flip rates of the knife-edge cases (tests/test_gpu_parity.py::KNIFE_EDGE_CASES).

The oracle is solved twice on the batch tests/test_gpu_parity_r2.py::test_knife_edge_flip_rate uses (B = 32, seed 20260929):
once as is, once with every sin / cos result moved by -1 / 0 / +1 ulp (oracle/models.hpp::trig_noise) -- the size of the
difference between glibc and the device libm.  The fraction of trajectories whose (status, iterations) change is the
yardstick for the HIP-vs-oracle flip rate: a GPU path that flipped MORE often than libm-level noise does would point at
an arithmetic difference, not at rounding.   python tests/golden/make_trig_noise.py  ->  tests/golden/trig_noise_flip_rates.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402


def flip_rates(api, cases=None):
    sys.path.insert(0, os.path.dirname(HERE))
    import test_gpu_parity as T
    lib = api.load_oracle()
    out = {}
    for case in sorted(cases or T.KNIFE_EDGE_CASES):
        p = T.TERM_CASES[case](api) if case in T.TERM_CASES else T.make(api, case)
        B = 32
        x0 = api.batch_x0(p, B, 20260929, T.spread_for(p))
        U0 = api.batch_U0(p, B)
        X0 = np.tile(p.X0_single, (B, 1, 1)) if hasattr(p, "X0_single") else None
        if X0 is not None:
            X0[:, 0, :] = x0
        lib.cddp_oracle_set_trig_noise(0)
        r0 = api.oracle_solve_batch(p, x0, U0, X0, n_threads=os.cpu_count() or 8, want_traj=False)[0]
        lib.cddp_oracle_set_trig_noise(1)
        try:
            r1 = api.oracle_solve_batch(p, x0, U0, X0, n_threads=os.cpu_count() or 8, want_traj=False)[0]
        finally:
            lib.cddp_oracle_set_trig_noise(0)
        lib.cddp_oracle_set_matmul_noise(1)      # second yardstick: <= 1 ulp on every matrix-product entry (summation order)
        try:
            r2 = api.oracle_solve_batch(p, x0, U0, X0, n_threads=os.cpu_count() or 8, want_traj=False)[0]
        finally:
            lib.cddp_oracle_set_matmul_noise(0)
        same = (r0["iterations"] == r1["iterations"]) & (r0["status"] == r1["status"])
        work = same & (r0["n_backward"] == r1["n_backward"]) & (r0["n_forward"] == r1["n_forward"])
        same2 = (r0["iterations"] == r2["iterations"]) & (r0["status"] == r2["status"])
        out[case] = {"B": B, "same_counts": int(same.sum()), "same_work": int(work.sum()),
                     "matmul_noise_same_counts": int(same2.sum()),
                     "converged_clean": int(np.sum((r0["status"] == 1) | (r0["status"] == 2)))}
    return out


if __name__ == "__main__":
    api = conftest.load_api()
    res = flip_rates(api)
    with open(os.path.join(HERE, "trig_noise_flip_rates.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))
