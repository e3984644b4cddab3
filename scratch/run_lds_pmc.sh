# LDS-array occupancy of the LDS-tiled kernels: separate --pmc passes (kernel-trace only), short bench run
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/ldspmc && mkdir -p gpurun_out/ldspmc
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/ldspmc/p$i -o run --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sequential > gpurun_out/ldspmc/p$i.log 2>&1 || { tail -5 gpurun_out/ldspmc/p$i.log; echo "pass $i ($set) failed"; }
done
python - <<'PY'
import csv, glob, collections, re, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ldspmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
        if "anonymous namespace" not in n or not m: continue
        if int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) < 200 * 1024: continue   # the 160-graph launches
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("gpurun_out/ldspmc/summary.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(out.items()): print(k, {c: round(v) for c, v in d.items()})
PY
rm -rf gpurun_out/ldspmc/p*/
