#!/bin/bash
# SQ / TCC counters of one workload's kernels (separate --pmc passes, kernel trace only).  usage: pmc_workload.sh <workload> [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=$1; shift
O=gpurun_out/pmc_$W; mkdir -p $O; rm -rf $O/*
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $W "$@" > $O/pmc_$tag.log 2>&1
done
python profiles/summarize_pmc.py $O/pmc_* > $O/pmc_counters.md
rm -rf $O/pmc_*/
cat $O/pmc_counters.md | cut -c1-330
