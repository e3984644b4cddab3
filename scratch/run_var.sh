# same-box timings of library variants: VARIANTS="hip t512 ..." (lib/libgcnmaxcut_<name>.so)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {
  GCN_MAXCUT_LIB=$2 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-sequential 2>>gpurun_out/var.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('$1', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, d.get('parity'))"
}
for rep in 1 2; do for v in ${VARIANTS:-hip t512}; do one $v $L/libgcnmaxcut_$v.so; done; done
