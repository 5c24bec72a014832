#!/bin/bash
# Round 3: the XCD-aware block map of the cooperative sweeps (kernels_coop.hpp::coop_group), A/B on the C4 / C5 shares:
# bench line + kernel trace + FETCH_SIZE / WRITE_SIZE counter passes (separate --pmc runs) with the map off and on.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/xcd_ab; mkdir -p $O; rm -rf $O/*
for w in quadrotor manip7; do
  for m in 0 1; do
    export CDDP_HIP_XCD_MAP=$m
    python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${w}_map$m.json
    rocprofv3 --kernel-trace --stats -d $O/trace_${w}_$m -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/trace_${w}_$m.log 2>&1
    python profiles/summarize_rocpd.py $O/trace_${w}_$m/r_results.db $O/kernel_stats_${w}_map$m.md | head -8
    for set in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${w}_${m}_$set -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/pmc_${w}_${m}_$set.log 2>&1
    done
    python profiles/summarize_pmc.py $O/pmc_${w}_${m}_* > $O/pmc_counters_${w}_map$m.md
    rm -rf $O/trace_${w}_$m $O/pmc_${w}_${m}_FETCH_SIZE $O/pmc_${w}_${m}_WRITE_SIZE
  done
done
unset CDDP_HIP_XCD_MAP
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/xcd_ab/bench_*.json')):
    d = json.load(open(f)); r = d['roofline']
    print(f.split('/')[-1], round(d['ms_per_step'], 1), 'ms', r['kernel'][:40], round(r['frac'], 3), {k: round(v['ms'], 1) for k, v in r['classes'].items() if isinstance(v, dict)})
PY
