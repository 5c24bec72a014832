#!/bin/bash
# round-1 build d: kernel trace + HBM traffic counters (separate passes) for the bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --stats -d gpurun_out/pmc/trace -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/trace.log 2>&1
for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/bench_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/bench_$tag.log 2>&1
done
python profiles/summarize_rocpd.py gpurun_out/pmc/trace/r_results.db
python profiles/summarize_pmc.py gpurun_out/pmc/bench_*
tail -1 gpurun_out/pmc/trace.log
