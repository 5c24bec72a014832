#!/usr/bin/env python3
"""Build the per-kernel HBM traffic JSON bench.py reads from a summarize_pmc.py table.
usage: make_traffic_json.py pmc_counters.md "<source note>" [outer_iterations_per_solve|0] [tile groups per solve] > rNN_x_pmc_traffic.json
bytes per launch = FETCH_SIZE x 1024 x 2 (gfx950 correction, profiles/r01_c_pmc_calibration.md) + WRITE_SIZE x 1024."""
import json
import sys

rows = [l.strip().strip("|").split("|") for l in open(sys.argv[1]) if l.startswith("|")]
hdr = [c.strip() for c in rows[0]]
fi, wi, di = hdr.index("FETCH_SIZE"), hdr.index("WRITE_SIZE"), hdr.index("dispatches")
# The number of solves inside the profiled command is READ from the table, not assumed: every solve starts with exactly one
# k_init launch (round 3 hard-coded 2 while `bench.py --steps 1 --warmup 0` had grown to three solves -- cold, profiling, timed --
# and the driver line's `traffic` was 1.5x too large).  With the optional third argument (outer iterations of one solve, e.g.
# the bench line's solve.max_iterations when no trajectory converges) the count is cross-checked against the sweep launches.
def _disp(prefix):
    return sum(int(r[di].strip()) for r in rows[2:] if r[0].strip().strip("`").startswith(prefix))
# round 5: a handle whose batch is cut into G concurrent tile groups (static CU partition, cddp_hip_num_groups) launches every kernel G
# times per iteration and k_init G times per solve: the fourth argument names G
GROUPS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if _disp("k_init") % GROUPS:
    raise SystemExit("k_init dispatches %d are not a multiple of %d tile groups" % (_disp("k_init"), GROUPS))
SOLVES = _disp("k_init") // GROUPS
if SOLVES <= 0:
    raise SystemExit("no k_init dispatches in %s: cannot tell how many solves the profiled command ran" % sys.argv[1])
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    sweeps = _disp("k_backward")
    if sweeps != SOLVES * GROUPS * int(sys.argv[3]):
        raise SystemExit("k_backward* dispatches %d != k_init dispatches %d x %s outer iterations" % (sweeps, SOLVES, sys.argv[3]))
kern = {}
for r in rows[2:]:
    c = [x.strip() for x in r]
    if not c[fi] or not c[wi]:
        continue
    f, w = float(c[fi]), float(c[wi])
    n = int(c[di])
    kern[c[0].strip("`")] = {"fetch_kb": f, "write_kb": w, "bytes_per_launch": f * 1024 * 2 + w * 1024, "dispatches": n,
                             "bytes_per_solve": (f * 1024 * 2 + w * 1024) * n / SOLVES}
print(json.dumps({
    "source": sys.argv[2],
    "calibration": "FETCH_SIZE reads 0.488x the known bytes of the 8-B/lane and 16-B/lane row streams of "
                   "profiles/ubench/vmem.hip on this gfx950 stack (guide: x2 correction); WRITE_SIZE reads 0.976x "
                   "the bytes of hipMemset fills (no correction). profiles/r01_c_pmc_calibration.md",
    "unit": "bytes per launch (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024); bytes_per_solve = that x dispatches / solves in the profiled command",
    "solves_in_profile": SOLVES, "tile_groups_per_solve": GROUPS,
    "kernels": kern}, indent=1))
