# co-residency experiment: n=400 graphs at FS=16 fit two 512-thread workgroups per CU
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
one() {  # name lib [env...]
  name=$1; lib=$2; shift 2
  env GCN_MAXCUT_LIB=$lib "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --nodes ${NODES:-400} --graphs-per-gpu ${GPG:-400} 2>>gpurun_out/t512.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('$name', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, d.get('parity'))"
}
for rep in 1 2; do
  one t1024_fs16 $L/libgcnmaxcut_hip.so GMC_LDS_MAX_FS=16
  one t512x2_fs16 $L/libgcnmaxcut_t512.so GMC_LDS_MAX_FS=16 GMC_DEVICE_CUS=512 GMC_DW1_CHUNKS=16
  one t512x1_fs16 $L/libgcnmaxcut_t512.so GMC_LDS_MAX_FS=16
  one t1024_fs32 $L/libgcnmaxcut_hip.so
done
