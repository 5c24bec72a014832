#!/bin/bash
# Round 4: GPU suite, HBM-traffic counters (separate --pmc passes) + kernel traces of the headline command and of the CLDDP / C4 / C5 /
# LogDDP workloads, the driver's bench line (with other_workloads; after the traffic JSONs so that its `traffic` fields are this
# build's), the 1-GPU batch curve, and the same-box A/B of the product against the device-libm comparison build.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O; rm -rf $O/*
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -40 > $O/gpu_suite.log; tail -3 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
prof() {   # prof <tag> <traffic json name> <bench args...>
  local tag=$1 tj=$2; shift 2
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads "$@" > $O/trace_$tag.log 2>&1
  python profiles/summarize_rocpd.py $O/trace_$tag/r_results.db $O/kernel_stats_$tag.md | head -9 | cut -c1-160
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${tag}_$set -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads "$@" > $O/pmc_${tag}_$set.log 2>&1
  done
  python profiles/summarize_pmc.py $O/pmc_${tag}_* > $O/pmc_counters_$tag.md
  python profiles/make_traffic_json.py $O/pmc_counters_$tag.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads $*, round 4 final build (profiles/scripts/final_r04.sh); per-kernel means in profiles/r04_pmc_counters_$tag.md" > $O/$tj && cp $O/$tj profiles/$tj
  rm -rf $O/trace_$tag $O/pmc_${tag}_*/
}
prof cartpole_ipddp r04_pmc_traffic.json --workload cartpole
prof cartpole_clddp r04_pmc_traffic_clddp.json --workload cartpole --solver clddp
prof quadrotor r04_pmc_traffic_quadrotor.json --workload quadrotor
prof manip7 r04_pmc_traffic_manip7.json --workload manip7
rocprofv3 --kernel-trace --stats -d $O/trace_logddp -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads --workload cartpole --solver logddp > $O/trace_logddp.log 2>&1
python profiles/summarize_rocpd.py $O/trace_logddp/r_results.db $O/kernel_stats_cartpole_logddp.md | head -8 | cut -c1-160; rm -rf $O/trace_logddp
rocprofv3 --kernel-trace --stats -d $O/trace_msipddp -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads --workload pendulum --solver msipddp > $O/trace_msipddp.log 2>&1
python profiles/summarize_rocpd.py $O/trace_msipddp/r_results.db $O/kernel_stats_pendulum_msipddp.md | head -8 | cut -c1-160; rm -rf $O/trace_msipddp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads > $O/pmc_sq.log 2>&1
python profiles/summarize_pmc.py $O/pmc_sq > $O/pmc_sq_cartpole_ipddp.md; rm -rf $O/pmc_sq
python bench.py --steps 10 --warmup 2 > $O/bench_cartpole_ipddp.json 2> $O/bench.err
for b in 1024 2048 4096 8192 16384 32768; do
  python bench.py --steps 4 --warmup 1 --batch $b --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/batch_curve.jsonl
done
# product (shared straight-line arithmetic) against the device-libm comparison build, same box, alternating
if [ -f cddp-cpp_amd/lib/libcddp_hip_libm.so ]; then
  one() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads $1 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],2))"; }
  for rep in 1 2; do for w in "--workload cartpole" "--workload cartpole --solver clddp" "--workload unicycle" "--workload quadrotor" "--workload manip7"; do
    echo "$w | shared: $(one "$w") | libm: $(CDDP_HIP_LIB=$GRAFT_REPO_ROOT/cddp-cpp_amd/lib/libcddp_hip_libm.so CDDP_HIP_TRIG=libm one "$w")"; done; done | tee $O/trig_ab.txt
fi
python - <<PY
import json
d=json.load(open('$O/bench_cartpole_ipddp.json')); print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline'])
for w in d.get('other_workloads', []): print({k: w[k] for k in w if k != 'roofline'}, w.get('roofline', {}).get('frac'), w.get('roofline', {}).get('traffic'))
for l in open('$O/batch_curve.jsonl'):
    d=json.loads(l); print(d['config'].get('batch_per_gpu'), round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3))
PY
