# same-box timings of library variants: VARIANTS="hip hip_base t512 ..." (lib/libgcnmaxcut_<name>.so), interleaved twice
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {
  GCN_MAXCUT_LIB=$2 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline 2>>gpurun_out/var.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; q=d.get('sequential') or {}
print('$1', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, 'seq us/graph-step', round(q.get('us_per_graph_step',0),2))"
}
for rep in 1 2; do for v in ${VARIANTS:-hip_base hip}; do one $v $L/libgcnmaxcut_$v.so; done; done
