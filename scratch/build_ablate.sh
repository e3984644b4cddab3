#!/bin/bash
# lib/libgcnmaxcut_ablN.so for N in "$@": the LDS-tiled TUs rebuilt with -DGMC_ABLATE=N, every other object reused
set -e
cd "$(dirname "$0")/../gcn-max-cut_amd/csrc"
make -s -j8
mkdir -p build_abl
for n in "$@"; do
  for f in spmm_lds fwd1_lds bwd1_lds; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DGMC_ABLATE=$n -c $f.hip -o build_abl/${f}_$n.o &
  done
done
wait
for n in "$@"; do
  objs=$(ls build/*.o | grep -v -E "/(spmm_lds|fwd1_lds|bwd1_lds).o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgcnmaxcut_abl$n.so $objs build_abl/spmm_lds_$n.o build_abl/fwd1_lds_$n.o build_abl/bwd1_lds_$n.o
done
ls ../lib
