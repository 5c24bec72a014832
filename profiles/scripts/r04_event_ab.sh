#!/bin/bash
# Round 4: class-timing events with / without the system-scope fence at the record (CDDP_HIP_EVENT_FENCE), stream loop vs graph replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_graph; mkdir -p $O
one() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],2), round(d['roofline'].get('frac',0),3))"; }
for rep in 1 2 3; do for w in "--workload cartpole" "--workload cartpole --solver clddp" "--workload unicycle" "--workload pendulum" "--workload quadrotor"; do
  echo "$w | fence: $(CDDP_HIP_EVENT_FENCE=1 one "$w") | nofence: $(CDDP_HIP_EVENT_FENCE=0 one "$w") | graph: $(CDDP_HIP_GRAPH=1 one "$w")"; done; done | tee $O/event_ab.txt
