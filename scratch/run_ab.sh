set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/ab.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms']); print(d['roofline']['achieved'], d['roofline']['mean_launch_us']); print(d.get('kernels_ms_unfused'))"
