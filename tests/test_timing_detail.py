"""cddp_hip_set_timing_detail: which kernel classes cddp_hip_solve brackets with hipEvents (include/cddp_hip.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_timing_detail_selects_the_bracketed_classes(api):
    p = api.cartpole_problem(api.SOLVER_IPDDP, True, 30)
    B = 64
    x0 = api.batch_x0(p, B, 20261011, 0.05 * np.ones(p.nx))
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0)
    ref = None
    for detail, want in ((api.TIMING_ROLLOUT, (False, True, False)), (api.TIMING_ALL, (True, True, True)),
                         (api.TIMING_SWEEP, (True, False, False))):
        hs.set_timing_detail(detail)
        st = hs.solve()
        got = (st.backward_ms > 0.0, st.forward_ms > 0.0, st.update_ms > 0.0)
        assert got == want and st.timing_detail == detail, (detail, got)
        assert st.backward_ms + st.forward_ms + st.update_ms <= st.solve_ms * 1.01
        res = hs.results()
        if ref is None:
            ref = res
        else:   # timing never changes the solve
            assert np.array_equal(ref["iterations"], res["iterations"]) and np.array_equal(ref["final_objective"], res["final_objective"])
    with pytest.raises(Exception):
        hs.set_timing_detail(7)
    hs.close()
