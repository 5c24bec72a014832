#!/bin/bash
# Bench lines of every workload without the CPU baseline (quick A/B of a kernel change).  usage: quick_bench.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out/quick_$1; mkdir -p $O; rm -f $O/*
for w in "cartpole" "cartpole --solver clddp" "unicycle" "quadrotor" "manip7" "pendulum"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 >> $O/bench.jsonl
done
python - <<PY
import json
for l in open("$O/bench.jsonl"):
    d = json.loads(l); c = d["roofline"]["classes"]
    print(d["config"]["workload"][:48].ljust(48), "ms/solve %8.2f" % d["ms_per_step"], "traj/s %9.0f" % d["value"], "frac %.3f" % d["roofline"]["frac"],
          "bwd %.1f fwd %.1f upd %.1f" % (c["backward(K1+K1b+K2+K3)"]["ms"], c["forward(K4)"]["ms"], c["update(K4b+K5)"]["ms"]))
PY
