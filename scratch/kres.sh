#!/bin/bash
# usage: scratch/kres.sh file.hip [filter-regex] -- per-kernel VGPR / scratch / occupancy summary
f=$1; pat=${2:-.}
cd "$(dirname "$0")/../gcn-max-cut_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | awk '/Function Name:/{n=$(NF-1)} / VGPRs:/{v=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} / SGPRs:/{sg=$(NF-1)} /LDS Size/{print n, "vgpr="v, "sgpr="sg, "scratch="s, "occ="o}' \
 | c++filt | sed 's/(anonymous namespace):://g; s/void //' | grep -E "$pat" | cut -c1-200
