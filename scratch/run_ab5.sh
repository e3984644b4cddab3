set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {
  GCN_MAXCUT_LIB=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sequential 2>>gpurun_out/ab5.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; u=d.get('kernels_ms_unfused') or {}
print('$1', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, 'spmm', round(d['roofline']['mean_launch_us'],1), {a:round(b*1000,1) for a,b in u.items() if a in ('agg_bwd','agg_fwd','dw1','gather_w1','hidden_bwd')})"
}
for v in hip_base hip hip_base hip; do one $v $L/libgcnmaxcut_$v.so; done
