#!/bin/bash
# Round 4: device-resident LogDDP -- parity tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_logddp; mkdir -p $O
timeout 900 python -m pytest tests/test_logddp_device.py -q -m gpu -n 4 2>&1 | tail -60 > $O/tests.log; tail -60 $O/tests.log
