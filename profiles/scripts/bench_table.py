"""The measurement table of DESIGN.md section 4 from one bench line.  usage: python profiles/scripts/bench_table.py profiles/r06_bench_cartpole_ipddp.json"""
import json, sys
d = json.load(open(sys.argv[1]))


def row(name, batch, ms, val, rf):
    alg, tr = rf.get("algorithmic_bytes_per_launch"), rf.get("traffic")
    wc = rf.get("whole_solve_counter_based") or {}
    wm = rf.get("whole_solve_frac")
    if wm is None: wm = (rf.get("classes", {}).get("whole_solve") or {}).get("frac")
    print("| %s | %s | %.2f | %.1f k | `%s` | **%.3f** | %s | %s | %s |" % (
        name, batch, ms, val / 1e3, rf.get("kernel", "").replace("+", "` + `"), rf["frac"], ("%.2f" % (tr / alg)) if alg and tr else "--",
        ("%.3f" % wm) if wm is not None else "--", ("%.3f" % wc["frac"]) if wc.get("frac") is not None else "--"))


print("| workload | batch | ms / solve | trajectories / s | dominant kernel class | frac of 8 TB/s | traffic / algorithmic | whole solve, 8(d) model | whole solve, counters |")
print("|---|---|---|---|---|---|---|---|---|")
row("C2 cart-pole IPDDP (headline, BASELINE config[1])", d["config"]["batch_per_gpu"], d["ms_per_step"], d["value"], d["roofline"])
sf, rest = [], []
for w in d.get("other_workloads", []):
    if w["workload"].startswith("stack-fed"): sf.append(w)
    elif w.get("roofline") and "classes_ms" in w: row(w["workload"].split(", B=")[0].split(" (")[0] if w["workload"].startswith("C") else w["workload"].split(", B=")[0].replace("f4: ", ""), w["batch"], w["ms_per_step"], w["value"], w["roofline"])
    else: rest.append(w)
print()
print("| stack-fed sweep (g1), one launch | branch | default form (c = cooperative, l = one lane per trajectory) | batch -> kernel ms, fraction of 8 TB/s by the 8(d) B_bwd bytes (every stack read, every gain row written ONCE) | counters of the best launch: traffic / algorithmic, HBM rate |")
print("|---|---|---|---|---|")
for w in sf:
    lab = w["workload"].split("): ")[1]
    shape, br = lab.rsplit(", ", 1) if "path rows" not in lab else (lab.rsplit(", ", 2)[0], "IPDDP, path rows")
    rf = w.get("roofline") or {}
    tr = ("%.2f x, %.2f TB/s (%.2f of peak)" % (rf["traffic_over_algorithmic"], rf["hbm_rate_GBps"] / 1e3, rf["hbm_frac"])) if rf.get("traffic") else "--"
    print("| %s | %s | %s | %s | %s |" % (shape, br, "c" if all(c.get("form") == "coop" for c in w["batch_curve"]) else ("l" if all(c.get("form") == "lane" for c in w["batch_curve"]) else "c up to %d, then l" % max(c["batch"] for c in w["batch_curve"] if c.get("form") == "coop")), ", ".join("%d: %.2f ms, %s%.3f%s" % (c["batch"], c["kernel_ms"], "**" if c is max(w["batch_curve"], key=lambda q: q.get("frac", 0)) else "", c["frac"], "**" if c is max(w["batch_curve"], key=lambda q: q.get("frac", 0)) else "") for c in w["batch_curve"] if "frac" in c), tr))
print()
for w in rest:
    if w["workload"].startswith("host plug-in"):
        ts = w["time_split"]
        print("Plug-in solve (g1): %s: **%.0f trajectories/s**, %.0f ms (host callbacks + line search %.0f ms on %d threads, GPU sections %.0f ms of which sweep kernels %.1f ms, %d batch sweeps); one host thread: %.0f trajectories/s\n" % (
            w["workload"].split("): ")[1], w["value"], w["ms_per_step"], ts["host_ms"], ts["threads"], ts["gpu_section_ms"], ts["sweep_kernel_ms"], ts["batch_sweeps"], w["single_thread"]["value"]))
    elif w["workload"].startswith("MPC"):
        p = w.get("provided_trajectory_warm_start", {})
        print("MPC re-solves (f1): %s: existing solver state %.0f re-solved trajectories/s, %.2f ms, %.1f iterations per re-solve; provided trajectory (shifted plan) %.0f /s, %.2f ms, %.1f iterations; cold %.2f ms, %.1f iterations; converged %s\n" % (
            w["workload"].split("): ")[1].split(", 8 shift")[0], w["value"], w["ms_per_step"], w["mean_iterations_per_resolve"], p.get("value", 0), p.get("ms_per_step", 0), p.get("mean_iterations_per_resolve", 0),
            w["cold"]["ms"], w["cold"]["mean_iterations"], w["converged_by_round"][-1]))
c = d["cpu_baseline"]
print("`cpu_baseline`: %.0f trajectories/s on %d threads (%s), single thread %.1f, scaling %.1f x; usable CPUs %s" % (c["value"], c["threads"], c["kind"], c["single_thread_value"], c["thread_scaling"], c["cpus"]["usable"]))
