#!/bin/bash
# Round 4: same-box A/B of the default (device libm) and the shared-arithmetic build on the five bench workloads.
cd $GRAFT_REPO_ROOT
O=gpurun_out/trig_ab; mkdir -p $O; rm -f $O/*
one() { python bench.py --steps $2 --warmup 1 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
 for trig in libm shared; do
  echo "$rep $trig C2-IPDDP $(CDDP_HIP_TRIG=$trig one '' 8)"
  echo "$rep $trig C2-CLDDP $(CDDP_HIP_TRIG=$trig one '--solver clddp' 8)"
  echo "$rep $trig C3 $(CDDP_HIP_TRIG=$trig one '--workload unicycle' 5)"
  echo "$rep $trig C4 $(CDDP_HIP_TRIG=$trig one '--workload quadrotor' 2)"
  echo "$rep $trig C5 $(CDDP_HIP_TRIG=$trig one '--workload manip7' 2)"
 done
done | tee $O/trig_ab.txt
