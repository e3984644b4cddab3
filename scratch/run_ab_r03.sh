# same-box A/B of library variants on the headline step: VARIANTS="old hip lead1 ..." (lib/libgcnmaxcut_<name>.so), twice
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {
  GCN_MAXCUT_LIB=$2 python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-other-workloads --no-dp-profile --no-spmm-sweep ${EXTRA} 2>>gpurun_out/ab.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; q=d.get('sequential') or {}
print('$1', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, 'seq us/graph-step', round(q.get('us_per_graph_step',0),2), 'loss', d['last_loss'])"
}
for rep in 1 2; do for v in ${VARIANTS:-old hip}; do one $v $L/libgcnmaxcut_$v.so; done; done
