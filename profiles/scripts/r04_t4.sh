#!/bin/bash
# Round 4: sub-tile-minor (T4) sweep-input stacks -- tests of the nx > 8 paths, A/B timing, PMC traffic of the C4 / C5 sweeps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_t4; mkdir -p $O; rm -rf $O/*
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -80 > $O/tests.log; tail -6 $O/tests.log
one() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in c.items() if isinstance(v,dict)})"; }
for t4 in 1 0 1 0; do echo "C4 t4=$t4 $(CDDP_HIP_T4=$t4 one '--workload quadrotor')"; echo "C5 t4=$t4 $(CDDP_HIP_T4=$t4 one '--workload manip7')"; done
for w in quadrotor manip7; do for t4 in 1 0; do
  for set in FETCH_SIZE WRITE_SIZE; do
    CDDP_HIP_T4=$t4 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcw_${w}_${t4}_$set -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/pmcw_${w}_${t4}_$set.log 2>&1
  done
  python profiles/summarize_pmc.py $O/pmcw_${w}_${t4}_* > $O/pmc_counters_${w}_t4_$t4.md
  grep -i "backward\|derivs\|condense" $O/pmc_counters_${w}_t4_$t4.md
  rm -rf $O/pmcw_${w}_${t4}_*/
done; done
