set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/ab.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms']); print(d['roofline'])"
GCN_MAXCUT_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/ab2.err | tail -1 | cut -c1-600
