#!/bin/bash
# scratch/variant.sh TU NAME "EXTRA FLAGS": compile one TU with extra flags and link a separate library scratch/libs/libNAME.so
cd /root/repo/cddp-cpp_amd/csrc
tu=$1; name=$2; extra=$3
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -fno-signed-zeros -DCDDP_TRIG_SHARED=1 $extra"
/opt/rocm/bin/hipcc $FLAGS -c $tu.hip -o /root/repo/scratch/objs/${tu}_$name.o 2> /tmp/variant_$name.log || { echo FAILED $name; grep error /tmp/variant_$name.log | head; exit 1; }
objs=$(ls ../build/*.o | grep -v "/$tu.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o /root/repo/scratch/libs/lib$name.so $objs /root/repo/scratch/objs/${tu}_$name.o -ldl && echo LINKED $name
