"""Turns the rocprofv3 outputs of a bench.py run into the small summaries kept under profiles/."""
import collections, csv, glob, json, re, sys
tag = sys.argv[1]  # e.g. r02a
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]
stats = (glob.glob(f"gpurun_out/prof_{tag}/*_kernel_stats.csv") + glob.glob(f"gpurun_out/prof_{tag}/*/*_kernel_stats.csv"))[0]
rows = [r for r in csv.DictReader(open(stats)) if "anonymous namespace" in r["Name"]]
with open(f"profiles/{rnd}_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "pct_of_gpu_time"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], f'{float(r["AverageNs"])/1e3:.2f}', f'{float(r["MinNs"])/1e3:.2f}',
                    f'{float(r["MaxNs"])/1e3:.2f}', r["Percentage"]])
pmc = collections.defaultdict(dict)
for ctr in ("fetch", "write"):
    f = (glob.glob(f"gpurun_out/pmc_{ctr}_{tag}/*_counter_collection.csv") + glob.glob(f"gpurun_out/pmc_{ctr}_{tag}/*/*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "anonymous namespace" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k][ctr.upper() + "_SIZE_KB_mean"] = sum(v) / len(v); pmc[k]["launches_" + ctr] = len(v)
out = {}
for k, v in pmc.items():
    fetch_kb, write_kb = v.get("FETCH_SIZE_KB_mean", 0.0), v.get("WRITE_SIZE_KB_mean", 0.0)
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
    # streaming reads (16 B/lane) -> doubled; WRITE_SIZE is exact for 16 B/lane stores. Units: KB.
    v["hbm_bytes_per_launch_corrected"] = int(2 * fetch_kb * 1024 + write_kb * 1024)
    out[k] = v
json.dump(out, open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
def pick(sub):
    for k, v in out.items():
        if sub in k:
            return v["hbm_bytes_per_launch_corrected"]
    return None
method = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --steps 5 --warmup 1 "
          "--no-cpu-baseline --no-sequential` (scratch/run_profile_r02.sh); KB units; FETCH_SIZE doubled (gfx950 wide coalesced reads, "
          f"MI355X_MICROARCH.md HBM section); mean over launches; see profiles/{rnd}_pmc_traffic.json")
fw, bw = pick("fwd1_lds_kernel"), pick("bwd1_reg_kernel") or pick("bwd1_lds_kernel")
json.dump({"rows": 160000, "F": 500, "fwd1_fused": fw, "bwd1_fused": bw, "method": method, "round": int(rnd[1:])},
          open("profiles/traffic_fused.json", "w"), indent=1)
plain, epi = pick("spmm_lds_kernel<16, 8, 4, false, false, false>"), pick("spmm_lds_kernel<16, 8, 4, true, false, false>")
if plain and epi:
    json.dump({"rows": 160000, "F": 500, "hbm_bytes_per_launch": (plain + epi) // 2,
               "detail": {"agg_bwd (plain SpMM)": plain, "agg_fwd (SpMM + fused H@W2)": epi, "fwd1 (fused forward)": fw, "bwd1 (fused backward)": bw},
               "method": method, "round": int(rnd[1:])}, open("profiles/traffic_spmm.json", "w"), indent=1)
for k, v in sorted(out.items()):
    print(k.ljust(60), {a: round(b) for a, b in v.items()})
