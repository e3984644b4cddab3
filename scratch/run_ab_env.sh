# same-box A/B of one environment switch: VAR=NAME VALS="0 1" [STEPS=200]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
one() {
  env $VAR=$1 python bench.py --steps ${STEPS:-200} --warmup 30 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; q=d.get('sequential') or {}
print('$VAR=$1', round(d['value'],1), round(d['ms_per_step'],4), {a:round(b*1000,1) for a,b in k.items()}, 'seq us/graph-step', round(q.get('us_per_graph_step',0),2), 'loss', d['last_loss'])"
}
for rep in 1 2 3; do for v in $VALS; do one $v || exit 1; done; done
