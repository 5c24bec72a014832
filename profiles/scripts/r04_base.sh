#!/bin/bash
# Round 4 re-entry baseline: full GPU suite on the HEAD build, then the quick bench lines of every workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_base; mkdir -p $O; rm -rf $O/*
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -60 > $O/tests.log; tail -6 $O/tests.log
bash profiles/scripts/quick_bench.sh r04base
