#!/bin/bash
# usage: scratch/build_variant.sh NAME "DEFS"  ->  lib/libgcnmaxcut_NAME.so with the LDS-tiled TUs, head.hip and finish.hip rebuilt with DEFS
set -e
name=$1; defs=$2
cd "$(dirname "$0")/../gcn-max-cut_amd/csrc"
make -s -j8
mkdir -p build_var
for f in spmm_lds fwd1_lds bwd1_lds head finish; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $defs -c $f.hip -o build_var/${f}_$name.o &
done
wait
objs=$(ls build/*.o | grep -v -E "/(spmm_lds|fwd1_lds|bwd1_lds|head|finish).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgcnmaxcut_$name.so $objs build_var/spmm_lds_$name.o build_var/fwd1_lds_$name.o build_var/bwd1_lds_$name.o build_var/head_$name.o build_var/finish_$name.o
ls -la ../lib/libgcnmaxcut_$name.so
