# same-box A/B: base library (HEAD~) vs candidate (working tree), interleaved twice
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
one() {
  GCN_MAXCUT_LIB=$1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; u=d.get('kernels_ms_unfused') or {}
print('$2', round(d['value'],1), d['ms_per_step'].__round__(4), {a:round(b*1000,1) for a,b in k.items()}, 'spmm', round(d['roofline']['mean_launch_us'],1), {a:round(b*1000,1) for a,b in u.items() if a in ('agg_bwd','agg_fwd','dw1','gather_w1')})"
}
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
for rep in 1 2; do
  for v in ${AB_VARIANTS:-base hip}; do
    f=$L/libgcnmaxcut_$v.so; [ $v = hip ] && f=$L/libgcnmaxcut_hip.so; [ $v = base ] && f=$L/libgcnmaxcut_hip_base.so
    one $f $v
  done
done
