#!/bin/bash
# build the csrc of a git revision (default HEAD) as lib/libgcnmaxcut_hip_base.so for same-box A/B runs (scratch/run_ab.sh)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" gcn-max-cut_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/gcn-max-cut_amd/csrc" -j8 -s OUT="$root/gcn-max-cut_amd/lib/libgcnmaxcut_hip_base.so"
rm -rf "$tmp"
ls -la "$root/gcn-max-cut_amd/lib/"
