# Line-search launch shape at the headline workload: the adaptive ladder (default) against a pinned first stage of k1 step sizes
# (CDDP_HIP_LS_FIRST) and the whole ladder in one launch (CDDP_HIP_LS_STAGES=1).  Same selected trials in every shape.
cd $GRAFT_REPO_ROOT
CDDP_HIP_DEBUG_LADDER=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads 2>&1 | grep ladder | tail -22
one() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads $W 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), d['solve']['rollouts_launched'], d['solve']['kernel_launches'], {k:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
for W in "" "--workload unicycle"; do
  echo "== ${W:-cartpole}"
  echo "adaptive  $(one)"
  for k in 2 4 6 8; do echo "k1=$k      $(CDDP_HIP_LS_FIRST=$k one)"; done
  echo "one-stage $(CDDP_HIP_LS_STAGES=1 one)"
done
