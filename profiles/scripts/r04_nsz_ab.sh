#!/bin/bash
# Round 4: -fno-signed-zeros (the product) against the previous build (libcddp_hip_vsz.so = signed zeros kept), all bench workloads, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_nsz; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -12 > $O/gpu_suite.log; tail -4 $O/gpu_suite.log
one() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), {k[:3]:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
for rep in 1 2; do for w in "--workload cartpole" "--workload cartpole --solver clddp" "--workload cartpole --solver logddp" "--workload unicycle" "--workload quadrotor" "--workload manip7" "--workload pendulum"; do
  echo "$w | nsz: $(one "$w") | signed zeros: $(CDDP_HIP_LIB=$GRAFT_REPO_ROOT/cddp-cpp_amd/lib/libcddp_hip_vsz.so one "$w")"; done; done | tee $O/ab.txt
