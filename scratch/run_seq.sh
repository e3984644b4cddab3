set -e
cd $GRAFT_REPO_ROOT
for per in 4 2 1; do
  GMC_LDS_SLICES_PER_WG=$per python bench.py --mode sequential --graphs-per-gpu 20 --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/seq_$per.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('per $per', round(d['ms_per_step'],4), 'eager', round(d['ms_per_step_eager_probed'],3), {a:round(b*1000,1) for a,b in k.items()})"
done
