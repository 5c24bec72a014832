#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_step3; mkdir -p $O; rm -rf $O/*
python -m pytest tests/test_failure_isolation.py tests/test_host_cpp.py tests/test_reference_quadrotor.py tests/test_shared_trig_parity.py tests/test_gpu_parity_r2.py -q -m gpu -n 4 2>&1 | tail -60 > $O/tests.log
one() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in c.items() if isinstance(v,dict)})"; }
for g in 1 2 3 4; do echo "CLDDP groups=$g $(CDDP_HIP_GROUPS=$g one '--solver clddp')"; done
for g in 1 2; do echo "pendulum groups=$g $(CDDP_HIP_GROUPS=$g one '--workload pendulum')"; done
for g in 1 2; do echo "C2 ipddp groups=$g $(CDDP_HIP_GROUPS=$g one '')"; done
for g in 1 2 4; do echo "C4 groups=$g $(CDDP_HIP_GROUPS=$g one '--workload quadrotor --steps 2')"; done
for g in 1 2 4; do echo "C5 groups=$g $(CDDP_HIP_GROUPS=$g one '--workload manip7 --steps 2')"; done
tail -25 $O/tests.log
