cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g; rm -rf gpurun_out/g/*
rocprofv3 --kernel-trace --stats -d gpurun_out/g/t5 -o r -- python bench.py --steps 1 --warmup 0 --workload manip7 > gpurun_out/g/bench_manip7.json 2> gpurun_out/g/err5.log
python profiles/summarize_rocpd.py gpurun_out/g/t5/r_results.db gpurun_out/g/kernel_stats_manip7.md | head -3
rocprofv3 --kernel-trace --stats -d gpurun_out/g/t4 -o r -- python bench.py --steps 1 --warmup 0 --workload quadrotor --batch 2048 > gpurun_out/g/bench_quadrotor.json 2> gpurun_out/g/err4.log
python profiles/summarize_rocpd.py gpurun_out/g/t4/r_results.db gpurun_out/g/kernel_stats_quadrotor.md | head -3
python bench.py --steps 3 --warmup 1 --workload unicycle --batch 8192 > gpurun_out/g/bench_unicycle.json 2>> gpurun_out/g/err4.log
for f in manip7 quadrotor unicycle; do python -c "
import json,sys;d=json.load(open('gpurun_out/g/bench_$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'])"; done
