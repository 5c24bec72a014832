cd $GRAFT_REPO_ROOT
W="--workload quadrotor"
CDDP_HIP_DEBUG_LADDER=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads $W 2>&1 | grep ladder | awk 'NR%3==1' | tail -14
one() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads $W 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), d['solve']['rollouts_launched'], d['solve']['kernel_launches'], {k:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
echo "adaptive  $(one)"
for k in 2 3 4 6 8; do echo "k1=$k      $(CDDP_HIP_LS_FIRST=$k one)"; done
echo "one-stage $(CDDP_HIP_LS_STAGES=1 one)"
