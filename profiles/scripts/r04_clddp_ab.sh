#!/bin/bash
# Round 4: CLDDP core after a kernel change -- the CLDDP parity tests, then the bench line (twice) and the per-kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_clddp_$1; mkdir -p $O; rm -rf $O/*
python -m pytest tests -q -m gpu -n 4 -k "clddp or bitwise or whole_batch or full_batch" 2>&1 | tail -15 > $O/tests.log; tail -4 $O/tests.log
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload cartpole --solver clddp"
rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $CMD > $O/trace.log 2>&1
python profiles/summarize_rocpd.py $O/trace/r_results.db $O/kernel_stats_clddp.md | head -7 | cut -c1-150
rm -rf $O/trace
for i in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload cartpole --solver clddp 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in c.items() if isinstance(v,dict)})"; done
