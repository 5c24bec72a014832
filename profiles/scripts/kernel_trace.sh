cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
for w in cartpole cartpole_unc; do
rocprofv3 --kernel-trace --stats -d gpurun_out/trace/$w -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $w > gpurun_out/trace/$w.log 2>&1
python profiles/summarize_rocpd.py gpurun_out/trace/$w/r_results.db | head -8
done
