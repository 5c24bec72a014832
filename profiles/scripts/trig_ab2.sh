#!/bin/bash
# why is the shared-arithmetic build 2x slower at C4 / C5?  work counters + kernel traces of both builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trig_ab2; mkdir -p $O; rm -rf $O/*
for w in quadrotor manip7; do for trig in libm shared; do
  CDDP_HIP_TRIG=$trig python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w $trig', round(d['ms_per_step'],1), d['solve'], {k:(round(v['ms'],1) if isinstance(v,dict) else v) for k,v in d['roofline']['classes'].items()})"
  CDDP_HIP_TRIG=$trig rocprofv3 --kernel-trace --stats -d $O/tr_${w}_$trig -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/tr_${w}_$trig.log 2>&1
  python profiles/summarize_rocpd.py $O/tr_${w}_$trig/r_results.db $O/kernel_stats_${w}_$trig.md | head -14
  rm -rf $O/tr_${w}_$trig
done; done 2>&1 | tee $O/out.txt
