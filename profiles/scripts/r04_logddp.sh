#!/bin/bash
# Round 4: device-resident LogDDP -- parity tests (resident + host routes, C++ mirror), then the batched throughput line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_logddp; mkdir -p $O
timeout 1200 python -m pytest tests/test_logddp_device.py tests/test_logddp.py tests/test_host_cpp.py tests/test_logddp_stack_fed.py -q -m gpu -n 4 2>&1 | tail -40 > $O/tests.log; tail -40 $O/tests.log
for i in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --workload cartpole --solver logddp 2>$O/bench.err | tail -1 > $O/bench_logddp.json; python -c "
import json,sys;d=json.load(open('$O/bench_logddp.json'));c=d['roofline']['classes'];print(round(d['ms_per_step'],2), round(d['value']), d['solve']['status'], d['solve']['mean_iterations'], {k:round(v['ms'],1) for k,v in c.items() if isinstance(v,dict)})"; done
tail -3 $O/bench.err
