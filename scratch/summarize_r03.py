"""rocprofv3 outputs of scratch/run_profile_r03.sh -> the summaries kept under profiles/ (run ON the GPU box; the
raw traces stay in gpurun_out/).  Everything is grouped by (kernel, workgroups): the 160-graph step's launches are not
averaged with the one-graph launches of the reference-schedule region or with the SpMM batch sweep's."""
import collections, csv, glob, json, os, re, subprocess, sys

tag = sys.argv[1]
rnd = "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # csrc_fingerprint


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def find(d, pat):
    return (glob.glob(f"gpurun_out/{d}/*{pat}") + glob.glob(f"gpurun_out/{d}/*/*{pat}"))[0]


def num(r, *names):
    for nm in names:
        if r.get(nm) not in (None, ""):
            return int(float(r[nm]))
    return 1


def workgroups(r):
    return (num(r, "Grid_Size_X", "Grid_Size") // max(1, num(r, "Workgroup_Size_X", "Workgroup_Size"))) * \
           (num(r, "Grid_Size_Y") // max(1, num(r, "Workgroup_Size_Y")))


# ---- kernel durations of the default bench command
agg = collections.defaultdict(list)
for r in csv.DictReader(open(find(f"prof_{tag}", "_kernel_trace.csv"))):
    name = r["Kernel_Name"]
    if "anonymous namespace" not in name and "_kernel" not in name:
        continue
    agg[(short(name), workgroups(r))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"profiles/{rnd}_bench_kernel_stats_by_grid.csv", "w") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "workgroups", "calls", "avg_us", "min_us", "max_us", "total_ms"])
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, g, len(v), f"{sum(v)/len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}", f"{sum(v)/1e3:.2f}"])
# the headline's timed region is the LAST thing the process does: its K steps are the last K launches of each of the
# four step kernels (the 256-workgroup backward is also launched by the data-parallel profile's 80-graph shards, so
# the by-grid table mixes two batch sizes in that row)
line = [l for l in open(f"gpurun_out/prof_{tag}/bench_line.json") if l.startswith("{")][-1]
K = json.loads(line)["steps"]
rows_t = [r for r in csv.DictReader(open(find(f"prof_{tag}", "_kernel_trace.csv")))]
rows_t.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(f"profiles/{rnd}_bench_kernel_stats_timed_region.csv", "w") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "workgroups", "calls (= --steps: the last K launches of the process)", "avg_us", "min_us", "max_us"])
    # the headline's launch shapes: forward / backward on 256 workgroups (NS = 7 flavours), head on one workgroup per
    # graph, fold + Adam on 489 - the parity gate's one-graph forwards behind the timed region have other grids
    bl = json.loads(line)
    B = bl["config"]["graphs_per_gpu"]
    shapes = (("fwd1_lds_kernel<16, 8, 4, false, 7, false>", 256), ("head_kernel", B),
              ("bwd1_reg_kernel<16, 4, false, 7, false, false>", 256), ("finish_kernel", 489))
    step_rows = []
    exact = {}   # the timed region's instantiation of each kernel (the reference-schedule region launches finish_kernel<1> on the same grid)
    for key, wgs in shapes:
        mine = [r for r in rows_t if key in r["Kernel_Name"] and workgroups(r) == wgs][-K:]
        exact[key] = mine[-1]["Kernel_Name"]
        step_rows += mine
        v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in mine]
        w.writerow([short(mine[-1]["Kernel_Name"]), wgs, len(v), f"{sum(v)/len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}"])
    step_rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    span = (int(step_rows[-1]["End_Timestamp"]) - int(step_rows[0]["Start_Timestamp"])) / 1e3
    # the launches bench.py's HIP events bracket (`kernels_ms`, `roofline`): the probe region = the FIRST launches of
    # these shapes in the process, 150 untimed steps then K probed ones
    for key, wgs in shapes:
        mine = [r for r in rows_t if r["Kernel_Name"] == exact[key] and workgroups(r) == wgs][150:150 + K]
        v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in mine]
        tag_ = {"fwd1": "fwd1_fused", "head": "head", "bwd1": "bwd1_fused", "fini": "finish"}[key[:4]]
        w.writerow([short(mine[-1]["Kernel_Name"]) + " - the K launches of the probe region (HIP events of the same run: "
                    f"{1e3 * bl['kernels_ms'][tag_]:.2f} us)", wgs, len(v), f"{sum(v)/len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}"])
    w.writerow(["(device time from the first forward to the last fold+Adam of those K steps, us per step)", "", K, f"{span / K:.2f}", "", ""])
open(f"profiles/{rnd}_bench_line_under_rocprof.json", "w").write(line)
bench_line = json.loads(line)

# ---- PMC traffic per (kernel, workgroups): FETCH_SIZE doubled (gfx950: wide coalesced reads are tallied at half,
# MI355X_MICROARCH.md, HBM section), WRITE_SIZE as is; both in KB
pmc = collections.defaultdict(dict)
for ctr in ("fetch", "write"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(find(f"pmc_{ctr}_{tag}", "_counter_collection.csv"))):
        name = r["Kernel_Name"]
        if "anonymous namespace" not in name and "_kernel" not in name:
            continue
        per[(short(name), workgroups(r))].append(float(r["Counter_Value"]))
    for k, v in per.items():
        pmc[k][ctr.upper() + "_SIZE_KB_mean"] = sum(v) / len(v)
        pmc[k]["launches_" + ctr] = len(v)
sha = bench.csrc_fingerprint()
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None
out = {}
for (k, g), v in pmc.items():
    v["hbm_bytes_per_launch_corrected"] = int(2 * v.get("FETCH_SIZE_KB_mean", 0.0) * 1024 + v.get("WRITE_SIZE_KB_mean", 0.0) * 1024)
    out[f"{k} @ {g} workgroups"] = v
method = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `python bench.py --steps 5 --warmup 1 "
          "--no-cpu-baseline --no-sequential --no-other-workloads --no-dp-profile` (scratch/run_profile_r03.sh); KB units; FETCH_SIZE "
          "doubled (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section); mean over launches of one (kernel, grid)")
json.dump({"csrc_sha": sha, "git_head_of_the_snapshot": head, "method": method, "kernels": out},
          open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1, sort_keys=True)


def pick(sub, wgs=None):
    best = None
    for (k, g), v in pmc.items():
        if sub in k and (wgs is None or g == wgs):
            if best is None or v.get("launches_fetch", 0) > best[1].get("launches_fetch", 0):
                best = ((k, g), v)
    return None if best is None else best[1]["hbm_bytes_per_launch_corrected"]


rows, F = 160000, 500
fw = pick("fwd1_lds_kernel", 256)
bw = pick("bwd1_reg_kernel", 256) or pick("bwd1_lds_kernel", 256)
json.dump({"rows": rows, "F": F, "fwd1_fused": fw, "bwd1_fused": bw, "csrc_sha": sha, "git_head_of_the_snapshot": head,
           "method": method, "round": 3}, open("profiles/traffic_fused.json", "w"), indent=1)
plain = pick("spmm_lds_kernel<16, 8, 4, false, false, false, 7>", 1280)
epi = pick("spmm_lds_kernel<16, 8, 4, true, false, false, 7>", 1280)
if plain and epi:
    json.dump({"rows": rows, "F": F, "hbm_bytes_per_launch": (plain + epi) // 2, "csrc_sha": sha, "git_head_of_the_snapshot": head,
               "detail": {"agg_bwd (plain SpMM)": plain, "agg_fwd (SpMM + fused H@W2)": epi}, "method": method, "round": 3},
              open("profiles/traffic_spmm.json", "w"), indent=1)
# SpMM batch sweep: durations from the bench line of the stats run, traffic from the PMC passes (same launches by grid)
sweep = {}
by_batch = (bench_line.get("roofline_spmm") or {}).get("by_batch") or {}
for B, rec in by_batch.items():
    wgs = rec["workgroups"]
    tp, te = pick("spmm_lds_kernel<16, 8, 4, false, false, false, 7>", wgs), pick("spmm_lds_kernel<16, 8, 4, true, false, false, 7>", wgs)
    tr = (tp + te) // 2 if tp and te else None
    sweep[B] = dict(rec, hbm_bytes_per_launch_pmc=tr, traffic_over_algorithmic=(tr / rec["algorithmic_bytes_per_launch"] if tr else None))
json.dump({"what": "spmm_lds_kernel (layer-1 aggregation, forward + backward launches of the one-kernel-per-operation leg) at B = 1 / 20 "
                   "/ 160 / 1024 graphs of n=1000 d=7, F=500 (SURVEY 8(d) C3); B = 1024 is the 160 graphs repeated (R = 1,024,000 rows)",
           "csrc_sha": sha, "method": method, "by_batch": sweep}, open(f"profiles/{rnd}_spmm_batch_sweep.json", "w"), indent=1)
for k, v in sorted(out.items()):
    print(k.ljust(70), {a: round(b) for a, b in v.items()})
print(open(f"profiles/{rnd}_bench_kernel_stats_timed_region.csv").read())
