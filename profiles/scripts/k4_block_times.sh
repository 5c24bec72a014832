cd $GRAFT_REPO_ROOT
export CDDP_HIP_LIB=$GRAFT_REPO_ROOT/cddp-cpp_amd/lib/libcddp_hip_time.so
python profiles/scripts/k4_block_times.py 4096
python profiles/scripts/k4_block_times.py 1024
python profiles/scripts/k4_block_times.py 4096 10
python profiles/scripts/k4_block_times.py 1024 10
