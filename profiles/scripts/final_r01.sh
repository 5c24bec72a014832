cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py --steps 5 --warmup 1 > gpurun_out/final/bench_cartpole_ipddp.json 2> gpurun_out/final/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/final/trace -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/trace.log 2>&1
for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/final/pmc_$tag -o r -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/pmc_$tag.log 2>&1
done
python profiles/summarize_rocpd.py gpurun_out/final/trace/r_results.db gpurun_out/final/kernel_stats.md | head -12
python profiles/summarize_pmc.py gpurun_out/final/pmc_* > gpurun_out/final/pmc_counters.md
grep "k_forward_ipddp_pc\|k_backward_ipddp_coop\|k_costate\|k_condense\|k_post\|kernel" gpurun_out/final/pmc_counters.md | cut -c1-200
for w in "cartpole --solver clddp" "cartpole_unc" "pendulum" "unicycle --batch 8192"; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 >> gpurun_out/final/bench_other.jsonl; done
tail -c 600 gpurun_out/final/bench_cartpole_ipddp.json
