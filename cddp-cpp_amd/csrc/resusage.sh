#!/bin/bash
# usage: resusage.sh file.hip  -> per-kernel VGPR / scratch / occupancy table
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name/ {n=$NF} /remark:.* VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /SGPRs:/ {sg=$(NF-1)} /LDS Size/ {printf "%-100s VGPR %4s AGPR %3s SGPR %4s scratch %6s occ %s\n", n, v, a, sg, s, o}' | sed 's/\[-Rpass[^ ]*//g'
