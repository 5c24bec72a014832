# Same-box A/B of library variants that differ in ONE translation unit (here: the cart-pole instantiation with other K4 prefetch
# distances, profiles/r03_k4_block_times.md section 3).  The variants are built in the build container before the gpurun call:
#   cd cddp-cpp_amd/csrc; mkdir -p ../build_v
#   for v in "1 1" "3 1" "1 2" "5 3"; do set -- $v
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -DCDDP_K4_DEPTH_P=$1 -DCDDP_K4_DEPTH_C=$2 \
#           -c inst_cartpole.hip -o ../build_v/inst_cartpole_$1$2.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcddp_hip_v$1$2.so $(ls ../build/*.o | grep -v inst_cartpole) ../build_v/inst_cartpole_$1$2.o -ldl
#   done
# (the per-wave clocks of k4_block_times.sh use the same recipe with -DCDDP_K4_TIMING -> libcddp_hip_time.so)
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['roofline']['classes'];print(round(d['ms_per_step'],2), {k:round(v['ms'],2) for k,v in c.items() if isinstance(v,dict)})"; }
for rep in 1 2; do for lib in cddp-cpp_amd/lib/libcddp_hip.so cddp-cpp_amd/lib/libcddp_hip_v*.so; do
  [ -f $lib ] && echo "$(basename $lib) $(CDDP_HIP_LIB=$GRAFT_REPO_ROOT/$lib one)"; done; done
