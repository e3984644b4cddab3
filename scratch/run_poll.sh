set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
one() {
  env "$@" python bench.py --no-cpu-baseline --no-sequential 2>>gpurun_out/poll.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('$*', round(d['value'],1), round(d['ms_per_step'],4), round(sum(k.values()),4))"
}
for rep in 1 2; do
one GCN_MAXCUT_POLL_LOSS=1
one GCN_MAXCUT_POLL_LOSS=0
done
python -m pytest tests -m gpu -q --timeout 900 -k "training or epoch or checkpoint or sequential or replayed or bench_line" 2>&1 | tail -3
