"""Step timeline of the default bench from a rocprofv3 --kernel-trace --memory-copy-trace run:
per replayed step, the gaps between its kernels, the loss copy, and the turnaround to the next step."""
import csv, glob, sys, statistics as st
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    tag = ("fwd1" if "fwd1_lds" in n else "head" if "head_kernel" in n else "bwd1" if "bwd1_" in n else
           "finish" if "finish_kernel" in n else None)
    if tag and int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0) >= 0:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag, int(r["Grid_Size"]) if "Grid_Size" in r else 0))
for f in mt:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?"), int(r.get("Size", 0) or 0)))
ev.sort()
# steps of the batched (160-graph) run: fwd1 with the big grid followed by head, bwd1, finish
big = max(g for s, e, t, g in ev if t == "fwd1")
rows = []
i = 0
while i < len(ev):
    s, e, t, g = ev[i]
    if t == "fwd1" and g == big:
        seq = [ev[i]]
        j = i + 1
        while j < len(ev) and len(seq) < 6 and not (ev[j][2] == "fwd1"):
            seq.append(ev[j]); j += 1
        tags = [x[2] for x in seq]
        if tags[:4] == ["fwd1", "head", "bwd1", "finish"]:
            nxt = next((x for x in ev[j:j + 3] if x[2] == "fwd1" and x[3] == big), None)
            rows.append((seq, nxt))
        i = j
    else:
        i += 1
def med(xs): return round(st.median(xs) / 1e3, 2) if xs else None
out = {}
out["steps"] = len(rows)
out["fwd1_us"] = med([q[0][1] - q[0][0] for q, n in rows]); out["head_us"] = med([q[1][1] - q[1][0] for q, n in rows])
out["bwd1_us"] = med([q[2][1] - q[2][0] for q, n in rows]); out["finish_us"] = med([q[3][1] - q[3][0] for q, n in rows])
out["gap_fwd_head"] = med([q[1][0] - q[0][1] for q, n in rows]); out["gap_head_bwd"] = med([q[2][0] - q[1][1] for q, n in rows])
out["gap_bwd_finish"] = med([q[3][0] - q[2][1] for q, n in rows])
cp = [(q[4][0] - q[3][1], q[4][1] - q[4][0]) for q, n in rows if len(q) > 4 and q[4][2].startswith("copy")]
out["gap_finish_copy"] = med([a for a, b in cp]); out["copy_us"] = med([b for a, b in cp])
out["turnaround_last_end_to_next_fwd"] = med([n[0] - q[-1][1] for q, n in rows if n and n[0] - q[-1][1] < 200000])
out["step_period"] = med([n[0] - q[0][0] for q, n in rows if n and n[0] - q[0][0] < 1000000])
print(out)
