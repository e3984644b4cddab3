#!/bin/bash
# lib/libgcnmaxcut_ablN.so for N in "$@": spmm_lds.hip rebuilt with -DGMC_ABLATE=N, every other object reused
set -e
cd "$(dirname "$0")/../gcn-max-cut_amd/csrc"
make -s -j8
for n in "$@"; do
  mkdir -p build_abl
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DGMC_ABLATE=$n -c spmm_lds.hip -o build_abl/spmm_lds_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls build/*.o | grep -v spmm_lds.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgcnmaxcut_abl$n.so $objs build_abl/spmm_lds_$n.o
done
ls ../lib
