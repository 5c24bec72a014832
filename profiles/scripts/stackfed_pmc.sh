#!/bin/bash
# SQ counters of the lane-cooperative stack-fed sweep (stacks_coop.hpp), separate --pmc passes.  usage: stackfed_pmc.sh nx nu m N B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/stackfed_pmc; mkdir -p $O; rm -rf $O/*
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -o r -- python profiles/scripts/stackfed_one.py "$@" coop > $O/pmc_$i.log 2>&1
done
python profiles/summarize_pmc.py $O/pmc_* > $O/counters.md
rm -rf $O/pmc_*/
cat $O/counters.md
