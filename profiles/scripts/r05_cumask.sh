#!/bin/bash
# Round 5: CU-partitioned streams (hipExtStreamCreateWithCUMask), VERDICT r04 item 1.  usage: r05_cumask.sh <tag> [bench.py workload args...]
# Every configuration: bench.py --steps 4 --warmup 1 without the CPU baseline / other workloads; one table row per configuration.
cd $GRAFT_REPO_ROOT; TAG=$1; shift; WL="$@"
O=gpurun_out/cumask_$TAG; mkdir -p $O; rm -f $O/*
run() {   # name, env assignments...
  local name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads $WL 2>$O/err_$name.log | tail -1 > $O/line_$name.json
  python - "$name" "$O/line_$name.json" "$*" <<'PY' | tee -a $O/table.md
import json, sys
name, path, envs = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(open(path).read()); c = d["roofline"]["classes"]
    print("| %s | `%s` | %.2f | %.0f | %.1f | %.1f | %.1f |" % (name, envs, d["ms_per_step"], d["value"], c["backward(K1+K1b+K2+K3)"]["ms"], c["forward(K4)"]["ms"], c["update(K4b+K5)"]["ms"]))
except Exception as e:
    print("| %s | `%s` | FAILED %s | | | | |" % (name, envs, e))
PY
}
echo "| configuration | environment | ms / solve | trajectories/s | sweep class ms | rollout class ms | update class ms |" | tee $O/table.md
echo "|---|---|---|---|---|---|---|" | tee -a $O/table.md
run base X=0
run hop_only_F CDDP_HIP_CUMASK=F=0-256
run hop_only_FW CDDP_HIP_CUMASK=F=0-256,W=0-256
run g2_free CDDP_HIP_GROUPS=2
run g2_halves CDDP_HIP_GROUPS=2 "CDDP_HIP_CUMASK=C=0-128|C=128-256"
run g2_pp CDDP_HIP_GROUPS=2 CDDP_HIP_PINGPONG=1
for sp in "F=0-176,W=176-208" "F=0-176,W=176-256,C=176-256" "F=0-192,W=192-256" "F=0-176,W=176-208,C=176-256" "F=0-208,W=208-256" "F=0-224,W=224-256" "F=0-192,W=192-256,C=128-256" "F=0-160,W=160-256,C=160-256" "F=0-128,W=128-256,C=128-256"; do
  n=$(echo $sp | tr '=,' '__')
  run g2_pp_$n CDDP_HIP_GROUPS=2 CDDP_HIP_PINGPONG=1 "CDDP_HIP_CUMASK=$sp"
  run g2_free_$n CDDP_HIP_GROUPS=2 "CDDP_HIP_CUMASK=$sp"
done
run g2_pp_ownW CDDP_HIP_GROUPS=2 CDDP_HIP_PINGPONG=1 "CDDP_HIP_CUMASK=F=0-192,W=192-224|F=0-192,W=224-256"
run g4_free_F176 CDDP_HIP_GROUPS=4 "CDDP_HIP_CUMASK=F=0-176,W=176-256"
run g4_quarters CDDP_HIP_GROUPS=4 "CDDP_HIP_CUMASK=C=0-64|C=64-128|C=128-192|C=192-256"
run base_again X=0
cat $O/table.md > gpurun_out/cumask_$TAG.md
