# in-place cost of each component of the fused kernels (GMC_ABLATE diagnostic builds; wrong results by construction)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {
  GCN_MAXCUT_LIB=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/abl.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('$1', round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()})"
}
one base $L/libgcnmaxcut_hip.so
for n in ${ABLS:-1 2 3 4 5 6 7}; do one abl$n $L/libgcnmaxcut_abl$n.so; done
one base $L/libgcnmaxcut_hip.so
