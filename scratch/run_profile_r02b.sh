# kernel stats of the DEFAULT bench command, launches grouped by (kernel, grid) so that the 160-graph step's launches
# are not averaged together with the one-graph launches of the `sequential` region
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02b}
mkdir -p gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- python bench.py > gpurun_out/prof_$TAG/bench_line.json 2> gpurun_out/prof_$TAG/err.txt
python - <<'PY' $TAG
import csv, glob, collections, re, sys
tag = sys.argv[1]
f = (glob.glob(f"gpurun_out/prof_{tag}/*_kernel_trace.csv") + glob.glob(f"gpurun_out/prof_{tag}/*/*_kernel_trace.csv"))[0]
agg = collections.defaultdict(list)
print(open(f).readline()[:400])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "anonymous namespace" not in name and "_kernel" not in name: continue
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    short = m.group(1) if m else name[:60]
    def num(*names):
        for nm in names:
            if r.get(nm) not in (None, ""):
                return int(float(r[nm]))
        return 1
    wg = (num("Grid_Size_X", "Grid_Size") // max(1, num("Workgroup_Size_X", "Workgroup_Size"))) * (num("Grid_Size_Y") // max(1, num("Workgroup_Size_Y")))
    agg[(short, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"gpurun_out/prof_{tag}/kernel_stats_by_grid.csv", "w") as out:
    w = csv.writer(out); w.writerow(["kernel", "workgroups", "calls", "avg_us", "min_us", "max_us", "total_ms"])
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, g, len(v), f"{sum(v)/len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}", f"{sum(v)/1e3:.2f}"])
print(open(f"gpurun_out/prof_{tag}/kernel_stats_by_grid.csv").read()[:3000])
PY
rm -f gpurun_out/prof_$TAG/run_kernel_trace.csv gpurun_out/prof_$TAG/*/run_kernel_trace.csv
tail -c 600 gpurun_out/prof_$TAG/bench_line.json
