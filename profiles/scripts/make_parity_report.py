"""Collect the parity_report_*.json files the -m gpu tests leave under gpurun_out/ into one table.
usage: python profiles/scripts/make_parity_report.py gpurun_out profiles/r02_parity_report.md"""
import glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(os.path.join(src, "parity_report_*.json"))):
    name = os.path.basename(f)[len("parity_report_"):-len(".json")]
    d = json.load(open(f))
    cells = []
    for k, v in d.items():
        if isinstance(v, list) and len(v) > 6:
            if all(isinstance(x, (int, float)) for x in v):
                v = "max %.3g (n=%d)" % (max(v), len(v)) if any(isinstance(x, float) for x in v) else "%d values, min %d max %d" % (len(v), min(v), max(v))
            else:
                v = "%d entries" % len(v)
        elif isinstance(v, float):
            v = "%.4g" % v
        elif isinstance(v, dict):
            v = ", ".join("%s: %s" % (a, ("%.3g" % b) if isinstance(b, float) else b) for a, b in v.items())
        elif isinstance(v, list):
            v = ", ".join(("%.3g" % x) if isinstance(x, float) else str(x) for x in v)
        cells.append("%s = %s" % (k, v))
    rows.append((name, "; ".join(cells)))
with open(dst, "w") as o:
    o.write("# Parity reports written by the `-m gpu` tests on MI355X (HIP path vs oracle), one row per report file\n\n")
    o.write("Produced by `python -m pytest tests -m gpu` (tests/test_gpu_parity_r2.py, test_full_size.py, test_mfma_sweep.py); "
            "`same_counts` = trajectories with identical (status, iterations), `same_work` = also identical sweep / rollout counts, "
            "`strict` = also objective within 1e-7; `oracle_*_noise_same_counts` = the oracle against itself with <= 1 ulp noise "
            "(the yardstick a flip rate is held to).\n\n| report | content |\n|---|---|\n")
    for n, c in rows: o.write("| `%s` | %s |\n" % (n, c))
print(len(rows), "reports ->", dst)
