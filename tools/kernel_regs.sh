#!/bin/bash
# VGPR / scratch / occupancy / LDS of every kernel in one translation unit:  tools/kernel_regs.sh tv_march_D [name-regex]
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I pytv-4d_amd/csrc -c "pytv-4d_amd/csrc/$1.hip" -o "/tmp/kr_$1.o" \
    -Rpass-analysis=kernel-resource-usage 2>&1 | awk -v f="${2:-.}" '
    /remark: Function Name:/ {name=$(NF-1)}
    /remark: +VGPRs:/ {v=$(NF-1)}
    /ScratchSize/ {s=$(NF-1)}
    /Occupancy/ {o=$(NF-1)}
    /LDS Size/ {l=$(NF-1); if (name ~ f) printf "vgpr %-4s scratch %-5s occ %-2s lds %-6s %s\n", v, s, o, l, name}'
