set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rl in 0 1; do
  rm -rf gpurun_out/confpmc$rl && mkdir -p gpurun_out/confpmc$rl
  GCN_MAXCUT_RELABEL=$rl rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d gpurun_out/confpmc$rl -o run --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sequential > gpurun_out/confpmc$rl.log 2>&1
  python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/confpmc$rl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]; m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
        if "anonymous namespace" not in n or not m: continue
        if int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) < 200 * 1024: continue
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if "fwd1" in k or "bwd1" in k: print("relabel=$rl", k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  rm -rf gpurun_out/confpmc$rl
done
