import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import conftest
api = conftest.load_api()
p = api.manipulator7_problem(api.SOLVER_IPDDP, 150, True, 16)
B = 2048
x0 = api.batch_x0(p, B, 1234, 0.02 * np.ones(p.nx))
hs = api.HipBatchSolver(p, B); hs.set_initial(x0); hs.initialize(); hs.backward(); hs.backward()
K, k = hs.gains()
a = k[::4, :16, 0]
m = a.mean(axis=0)
print("TE phases, cycles: init %d | P1 %d (%.0f per step) | P2 %d (%.0f per step) | P3 %d | P4 %d | P5 %d (%.0f per step) | total %d" % (m[0], m[1], m[1] / p.N, m[2], m[2] / p.N, m[3], m[4], m[5], m[5] / p.N, m.sum()))
print('P1 sections per step: round1 %d | round2a %d | factor+K %d | variant %d | round3 %d | storeAB+Vc %d | stores %d' % tuple(m[8:15] / p.N)[:7] if False else 'P1 per step: ' + ' | '.join('%d' % (x / p.N) for x in m[8:14]))
print('P3 parts: As %d | AtA %d | reg %d | scales %d | pick %d' % (m[6], m[7], m[14], m[15], m[3]))
hs.close()
