#!/bin/bash
# Round 4: hipGraph replay of the iteration windows -- bitwise test, then same-box A/B of ms per solve
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_graph; mkdir -p $O
timeout 600 python -m pytest tests/test_determinism.py -q -m gpu -k graph 2>&1 | tail -5
one() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],2), d['solve']['kernel_launches'])"; }
for rep in 1 2; do for w in "--workload cartpole" "--workload cartpole --solver clddp" "--workload unicycle" "--workload pendulum"; do
  echo "$w | stream: $(CDDP_HIP_GRAPH=0 one "$w") | graph: $(CDDP_HIP_GRAPH=1 one "$w")"; done; done | tee $O/ab.txt
