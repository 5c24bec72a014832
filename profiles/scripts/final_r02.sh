#!/bin/bash
# Round 2: bench lines of every workload, kernel trace and HBM-traffic counters (separate --pmc passes) of the headline command.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r02; mkdir -p $O; rm -rf $O/*
python bench.py --steps 10 --warmup 2 > $O/bench_cartpole_ipddp.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/trace.log 2>&1
for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$tag.log 2>&1
done
python profiles/summarize_rocpd.py $O/trace/r_results.db $O/kernel_stats_cartpole_ipddp.md | head -12
python profiles/summarize_pmc.py $O/pmc_* > $O/pmc_counters.md
python profiles/make_traffic_json.py $O/pmc_counters.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --steps 1 --warmup 0 --no-cpu-baseline, round 2 build (profiles/scripts/final_r02.sh); per-kernel means in profiles/r02_pmc_counters.md" > $O/pmc_traffic.json
for w in "cartpole --solver clddp" "unicycle" "quadrotor" "manip7" "pendulum"; do
  python bench.py --steps 3 --warmup 1 --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl
done
for w in quadrotor manip7 unicycle; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$w -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > $O/trace_$w.log 2>&1
  python profiles/summarize_rocpd.py $O/trace_$w/r_results.db $O/kernel_stats_$w.md | head -6
done
rm -rf $O/trace/*.db $O/trace_*/ $O/pmc_*/   # raw databases stay on the box; the summaries travel
python -c "
import json
d=json.load(open('$O/bench_cartpole_ipddp.json')); print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])
for l in open('$O/bench_other_workloads.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:50], round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), d['cpu_baseline']['value'] if d['cpu_baseline'] else None)
"
