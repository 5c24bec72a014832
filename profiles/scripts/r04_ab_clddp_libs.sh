#!/bin/bash
# Same-box A/B of library variants on the CLDDP core (cart-pole, B = 4096): the product against cddp-cpp_amd/lib/libcddp_hip_v*.so
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --workload cartpole --solver clddp 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), {k:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
for rep in 1 2 3; do for lib in cddp-cpp_amd/lib/libcddp_hip.so cddp-cpp_amd/lib/libcddp_hip_v*.so; do
  [ -f $lib ] && echo "$(basename $lib) $(CDDP_HIP_LIB=$GRAFT_REPO_ROOT/$lib one)"; done; done
