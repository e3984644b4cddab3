set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for per in 4 2; do
GMC_LDS_SLICES_PER_WG=$per python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/per.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('per $per', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()})"
done; done
