cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), {k:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
for rep in 1 2; do for v in "" _v11 _v31 _v12 _v53; do echo "lib$v $(CDDP_HIP_LIB=$GRAFT_REPO_ROOT/cddp-cpp_amd/lib/libcddp_hip$v.so one)"; done; done
