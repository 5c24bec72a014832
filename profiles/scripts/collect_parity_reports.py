#!/usr/bin/env python3
"""Collect gpurun_out/parity_report_<prefix>_*.json (written by tests/test_cross_arithmetic.py on the GPU box) into one committed summary.
usage: collect_parity_reports.py <prefix: cross | assoc> "<source note>" out.json"""
import glob, json, os, sys
prefix, note, out = sys.argv[1], sys.argv[2], sys.argv[3]
w = {}
for f in sorted(glob.glob(os.path.join("gpurun_out", "parity_report_%s_*.json" % prefix))):
    d = json.load(open(f))
    w[d["workload"]] = d
json.dump({"source": note, "workloads": w}, open(out, "w"), indent=1)
print(out, {k: (v["compared"], v["count_flip_frac"], v["work_flip_frac"]) for k, v in w.items()})
