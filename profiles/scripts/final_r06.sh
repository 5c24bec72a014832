#!/bin/bash
# Round 6: kernel traces + HBM-traffic counters (separate --pmc passes) of the headline command and of the CLDDP / C3 / C4 / C5 / LogDDP /
# MSIPDDP workloads (traffic JSONs first, so that the bench line's `traffic` fields are this build's), then the driver's bench line.
# usage: final_r06.sh [suite]   (suite: also run the whole -m gpu suite serially first)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r06; mkdir -p $O; rm -rf $O/*
if [ "$1" = "suite" ]; then python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/gpu_suite_serial.log; tail -3 $O/gpu_suite_serial.log; fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
prof() {   # prof <tag> <traffic json name> <tile groups per solve> <bench args...>
  local tag=$1 tj=$2 ng=$3; shift 3
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads "$@" > $O/trace_$tag.log 2>&1
  python profiles/summarize_rocpd.py $O/trace_$tag/r_results.db $O/kernel_stats_$tag.md | head -9 | cut -c1-160
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${tag}_$set -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads "$@" > $O/pmc_${tag}_$set.log 2>&1
  done
  python profiles/summarize_pmc.py $O/pmc_${tag}_* > $O/pmc_counters_$tag.md
  python profiles/make_traffic_json.py $O/pmc_counters_$tag.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads $*, round 6 build (profiles/scripts/final_r06.sh); per-kernel means in profiles/r06_pmc_counters_$tag.md" 0 $ng > $O/$tj && cp $O/$tj profiles/$tj
  rm -rf $O/trace_$tag $O/pmc_${tag}_*/
}
prof cartpole_ipddp r06_pmc_traffic.json 2 --workload cartpole
prof cartpole_clddp r06_pmc_traffic_clddp.json 2 --workload cartpole --solver clddp
prof unicycle r06_pmc_traffic_unicycle.json 2 --workload unicycle
prof quadrotor r06_pmc_traffic_quadrotor.json 2 --workload quadrotor
prof manip7 r06_pmc_traffic_manip7.json 2 --workload manip7
prof cartpole_logddp r06_pmc_traffic_logddp.json 1 --workload cartpole --solver logddp
prof pendulum_msipddp r06_pmc_traffic_msipddp.json 1 --workload pendulum --solver msipddp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads > $O/pmc_sq.log 2>&1
python profiles/summarize_pmc.py $O/pmc_sq > $O/pmc_sq_cartpole_ipddp.md; rm -rf $O/pmc_sq
python bench.py --steps 10 --warmup 2 > $O/bench_cartpole_ipddp.json 2> $O/bench.err
for b in 1024 2048 4096 8192 16384 32768; do
  python bench.py --steps 4 --warmup 1 --batch $b --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/batch_curve.jsonl
done
python - <<PY
import json
d=json.load(open('$O/bench_cartpole_ipddp.json')); print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), {k: v for k, v in d['cpu_baseline'].items() if k != 'builds'})
print('whole', d['roofline']['classes']['whole_solve'], d['roofline']['whole_solve_counter_based'])
for w in d.get('other_workloads', []): print({k: w[k] for k in w if k != 'roofline'}, w.get('roofline', {}).get('frac'), w.get('roofline', {}).get('traffic'), w.get('roofline', {}).get('whole_solve_counter_based'))
for l in open('$O/batch_curve.jsonl'):
    d=json.loads(l); print(d['config'].get('batch_per_gpu'), round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3))
PY
