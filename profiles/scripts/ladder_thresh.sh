cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads $W 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), d['solve']['rollouts_launched'], d['solve']['kernel_launches'], {k:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
for m in 512 640 768 1024 512; do echo "max=$m $(CDDP_HIP_LS_TWO_MAX_WAVES=$m one)"; done
W="--workload pendulum"; for m in 512 768 1024; do echo "pend max=$m $(CDDP_HIP_LS_TWO_MAX_WAVES=$m one)"; done
W="--workload cartpole --solver clddp"; for m in 512 768 1024; do echo "clddp max=$m $(CDDP_HIP_LS_TWO_MAX_WAVES=$m one)"; done
