set -e
cd $GRAFT_REPO_ROOT
for per in 4 2 1; do
GMC_LDS_SLICES_PER_WG=$per python bench.py --graphs-per-gpu 20 --nodes 500 --steps 200 --warmup 30 --no-cpu-baseline 2>>gpurun_out/cfg1.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('B=20 n=500 per $per', round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()})"
done
