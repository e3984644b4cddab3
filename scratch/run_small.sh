set -e
cd $GRAFT_REPO_ROOT
python bench.py --graphs-per-gpu 20 --nodes 500 --steps 200 --warmup 30 --no-cpu-baseline 2>>gpurun_out/small.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('B=20 n=500 batched', round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()})"
python bench.py --mode sequential --graphs-per-gpu 20 --steps 50 --warmup 10 --no-cpu-baseline 2>>gpurun_out/small.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('sequential 20 x n=1000', round(d['ms_per_step'],4), 'eager', round(d['ms_per_step_eager_probed'],3), {a:round(b*1000,1) for a,b in k.items()})"
python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>>gpurun_out/small.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('B=160 n=1000 batched', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, d['parity'])"
