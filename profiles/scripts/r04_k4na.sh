#!/bin/bash
# Round 4: rollout workgroups of NA producers + one consumer (kernels_pcm.hpp) against the two-wave form: parity tests, then A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_k4na; mkdir -p $O; rm -rf $O/*
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_full_size.py tests/test_determinism.py tests/test_option_branches.py tests/test_warm_start.py tests/test_mpc_resolve.py -q -m gpu -n 4 2>&1 | tail -30 > $O/tests.log; tail -5 $O/tests.log
one() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3), round(c['whole_solve']['frac'],3), {k:round(v['ms'],1) for k,v in c.items() if isinstance(v,dict)})"; }
for rep in 1 2; do for na in 1 2 3; do
  echo "C2 NA=$na $(CDDP_HIP_K4_NA=$na one '')"
  echo "C3 NA=$na $(CDDP_HIP_K4_NA=$na one '--workload unicycle')"
  echo "pend NA=$na $(CDDP_HIP_K4_NA=$na one '--workload pendulum')"
done; done | tee $O/ab.txt
