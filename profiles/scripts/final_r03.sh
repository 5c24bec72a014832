#!/bin/bash
# Round 3: the driver's bench line (with other_workloads), kernel trace + HBM-traffic counters (separate --pmc passes) of the
# headline command, kernel traces / counters of the C4 / C5 shares, FETCH_SIZE calibration on the 32-B pattern per configuration,
# and the 1-GPU batch curve of the headline workload.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r03; mkdir -p $O; rm -rf $O/*
python bench.py --steps 10 --warmup 2 > $O/bench_cartpole_ipddp.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads > $O/trace.log 2>&1
for set in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads > $O/pmc_$tag.log 2>&1
done
python profiles/summarize_rocpd.py $O/trace/r_results.db $O/kernel_stats_cartpole_ipddp.md | head -12
python profiles/summarize_pmc.py $O/pmc_* > $O/pmc_counters.md
python profiles/make_traffic_json.py $O/pmc_counters.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads, round 3 build (profiles/scripts/final_r03.sh); per-kernel means in profiles/r03_pmc_counters.md" > $O/pmc_traffic.json
rm -rf $O/trace $O/pmc_*/
for w in quadrotor manip7; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$w -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/trace_$w.log 2>&1
  python profiles/summarize_rocpd.py $O/trace_$w/r_results.db $O/kernel_stats_$w.md | head -6
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcw_${w}_$set -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/pmcw_${w}_$set.log 2>&1
  done
  python profiles/summarize_pmc.py $O/pmcw_${w}_* > $O/pmc_counters_$w.md
  rm -rf $O/trace_$w $O/pmcw_${w}_*/
done
# FETCH_SIZE calibration on the 32-B-of-512-B pattern, one line per (K, B, map) configuration
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w profiles/ubench/vmem32.hip -o $O/vmem32 && {
  $O/vmem32 > $O/vmem32.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cal -o r -- $O/vmem32 > $O/pmc_cal.log 2>&1
  python profiles/summarize_pmc.py $O/pmc_cal > $O/pmc_vmem32.md
  rm -rf $O/pmc_cal $O/vmem32
}
# batch curve, 1 GPU
for b in 1024 2048 4096 8192 16384 32768; do
  python bench.py --steps 5 --warmup 1 --batch $b --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/batch_curve.jsonl
done
python - <<PY
import json
d=json.load(open('$O/bench_cartpole_ipddp.json')); print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])
for w in d.get('other_workloads', []): print(w)
for l in open('$O/batch_curve.jsonl'):
    d=json.loads(l); print(d['config'].get('batch_per_gpu', d['config']), round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline'].get('whole_solve_frac'))
PY
cat $O/vmem32.txt $O/pmc_vmem32.md
