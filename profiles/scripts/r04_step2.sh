#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_step2; mkdir -p $O; rm -rf $O/*
python -m pytest tests/test_failure_isolation.py tests/test_host_cpp.py tests/test_reference_quadrotor.py tests/test_shared_trig_parity.py tests/test_determinism.py tests/test_full_size.py -q -m gpu -x 2>&1 | tail -60 > $O/tests.log
rocprofv3 --kernel-trace --stats -d $O/tr_clddp -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --solver clddp > $O/tr_clddp.log 2>&1
python profiles/summarize_rocpd.py $O/tr_clddp/r_results.db $O/kernel_stats_clddp.md | head -12
rm -rf $O/tr_clddp
for b in 4096 8192 16384 32768; do
  python bench.py --steps 3 --warmup 1 --batch $b --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/batch_curve.jsonl
done
for b in 16384; do
  CDDP_HIP_CHUNK=4096 python bench.py --steps 3 --warmup 1 --batch $b --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/batch_curve_chunk4096.jsonl
  CDDP_HIP_CHUNK=0 python bench.py --steps 3 --warmup 1 --batch $b --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/batch_curve_nochunk.jsonl
done
python - <<PY
import json
for f in ("batch_curve","batch_curve_chunk4096","batch_curve_nochunk"):
  for l in open("$O/%s.jsonl"%f):
    d=json.loads(l); c=d["roofline"]["classes"]; print(f, d["config"]["batch_per_gpu"], round(d["value"]), round(d["ms_per_step"],2), round(d["roofline"]["frac"],3), round(c["backward(K1+K1b+K2+K3)"]["ms"],1), round(c["forward(K4)"]["ms"],1), round(c["update(K4b+K5)"]["ms"],1))
PY
tail -30 $O/tests.log
