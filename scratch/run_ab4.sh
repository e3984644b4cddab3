set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "gradients or fused or step or weighted or smallest or kink or full_size" 2>&1 | tail -3
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {
  GCN_MAXCUT_LIB=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sequential 2>>gpurun_out/ab4.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('$1', round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()})"
}
for v in ${VARIANTS:-hip_base hip hip_base hip}; do one $v $L/libgcnmaxcut_$v.so; done
