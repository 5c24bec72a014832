#!/usr/bin/env python3
"""Round 5: static CU partitions -- the batch cut into n tile groups, each group's every kernel on its own slice of the chip
(CDDP_HIP_GROUPS=n + CDDP_HIP_CUMASK="C=...|C=...").  usage: r05_partition.py <tag> <configs,comma> -- <bench.py args>
configs: base, symN (N contiguous mask-bit ranges = N symmetric slices of every XCD), xcdN (N sets of whole XCDs), groupsN (no masks)."""
import json, os, subprocess, sys

def spec(cfg):
    if cfg == "base":
        return {}
    if cfg.startswith("groups"):
        return {"CDDP_HIP_GROUPS": cfg[6:]}
    if cfg.startswith("sym"):
        n = int(cfg[3:]); w = 256 // n
        return {"CDDP_HIP_GROUPS": str(n), "CDDP_HIP_CUMASK": "|".join("C=%d-%d" % (k * w, (k + 1) * w) for k in range(n))}
    if cfg.startswith("xcd"):
        n = int(cfg[3:]); w = 8 // n
        return {"CDDP_HIP_GROUPS": str(n), "CDDP_HIP_CUMASK": "|".join("C=x" + "".join(str(k * w + j) for j in range(w)) for k in range(n))}
    raise SystemExit("unknown config " + cfg)

def main():
    tag, cfgs = sys.argv[1], sys.argv[2].split(",")
    args = sys.argv[sys.argv.index("--") + 1:]
    root = os.environ.get("GRAFT_REPO_ROOT", ".")
    out = os.path.join(root, "gpurun_out", "partition_%s.md" % tag)
    rows = ["| configuration | environment | ms / solve | trajectories/s | sweep class ms | rollout class ms | update class ms |", "|---|---|---|---|---|---|---|"]
    for cfg in cfgs:
        env = dict(os.environ); e = spec(cfg); env.update(e)
        try:
            r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-other-workloads"] + args,
                               env=env, capture_output=True, text=True, timeout=240)
            d = json.loads(r.stdout.strip().splitlines()[-1]); c = d["roofline"]["classes"]
            rows.append("| %s | `%s` | %.2f | %.0f | %.1f | %.1f | %.1f |" % (cfg, " ".join("%s=%s" % kv for kv in e.items()), d["ms_per_step"], d["value"],
                        c["backward(K1+K1b+K2+K3)"]["ms"], c["forward(K4)"]["ms"], c["update(K4b+K5)"]["ms"]))
        except subprocess.TimeoutExpired:
            rows.append("| %s | | TIMEOUT (240 s) | | | | |" % cfg)
        except Exception as ex:
            rows.append("| %s | | FAILED %s | | | | |" % (cfg, str(ex)[:80]))
        print(rows[-1], flush=True)
        open(out, "w").write("\n".join(rows) + "\n")

main()
