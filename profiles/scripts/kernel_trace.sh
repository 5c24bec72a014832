# Per-kernel durations of one workload each (rocprofv3 --kernel-trace --stats).  usage: kernel_trace.sh <tag> <workload> [<workload> ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift; O=gpurun_out/trace_$T; mkdir -p $O
for w in "$@"; do
  rocprofv3 --kernel-trace --stats -d $O/$w -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload $w > $O/$w.log 2>&1
  echo "== $w"; python profiles/summarize_rocpd.py $O/$w/r_results.db | head -12 | cut -c1-170
done
