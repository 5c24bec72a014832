#!/bin/bash
# FETCH_SIZE on the 32-B-of-512-B access pattern of the G = 16 sweeps (profiles/ubench/vmem32.hip), XCD map off / on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/calib32; mkdir -p $O; rm -rf $O/*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w profiles/ubench/vmem32.hip -o $O/vmem32 || exit 1
$O/vmem32 > $O/vmem32.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH -o r -- $O/vmem32 > $O/pmc_fetch.log 2>&1
python profiles/summarize_pmc.py $O/pmc_FETCH > $O/pmc_vmem32.md
rm -rf $O/pmc_FETCH $O/vmem32
cat $O/vmem32.txt $O/pmc_vmem32.md
