#!/bin/bash
# Round 4: per-kernel durations + SQ instruction counters of the CLDDP core at BASELINE config[1] (cart-pole, control box, B = 4096)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_clddp${1:+_$1}; mkdir -p $O; rm -rf $O/*
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload cartpole --solver clddp"
rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $CMD > $O/trace.log 2>&1
python profiles/summarize_rocpd.py $O/trace/r_results.db $O/kernel_stats_clddp.md | cut -c1-170
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o r -- $CMD > $O/pmc_sq.log 2>&1
python profiles/summarize_pmc.py $O/pmc_sq > $O/pmc_counters_clddp.md; cat $O/pmc_counters_clddp.md | cut -c1-220
rm -rf $O/trace $O/pmc_sq
for i in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload cartpole --solver clddp 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in c.items() if isinstance(v,dict)})"; done
