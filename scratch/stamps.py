import sys, os, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GCN_MAXCUT_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gcn-max-cut_amd", "lib", "libgcnmaxcut_hip_stamp.so")
import ctypes as C, numpy as np, torch, networkx as nx
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.Training import TrainingNeural as T
B, n, d, F = 160, 1000, 7, 500
# MODE=d8: 8-regular graphs; MODE=hub: d = 7 with ONE degree-40 hub in the last graph (the OVF flavours of the kernels)
MODE = os.environ.get("MODE", "d7")
def graph(i):
    if MODE == "gnp":   # G(n, p = 10/n) as bench.py's other_workloads
        import bench
        return bench.gnp_graph(n, 0.01, 200 + i)
    if MODE == "gnp16":  # the same graphs with every degree capped at 16 (timing only)
        import bench
        g = bench.gnp_graph(n, 0.01, 200 + i).copy()
        for v in list(g.nodes()):
            while g.degree(v) > 16:
                g.remove_edge(v, max(g.neighbors(v), key=g.degree))
        return g
    g = nx.random_regular_graph(8 if MODE == "d8" else d, n, seed=3000 + i)
    if MODE == "hub" and i == B - 1:
        for v in range(10, 10 + 33): g.add_edge(5, v)
    return g
hs = [pkg.from_networkx(graph(i)) for i in range(B)]
batch = pkg.GraphBatch(hs, None)
net, _, _ = T.setup_model_and_optimizer(T.TrainingConfig(n_nodes=1000, hidden_dim=F))
eng = net.engine(); lib = pkg.hip.load()
names = {"fwd": ["loop", "dma_wait", "barrier1", "gather1", "barrier2", "stores+dma", "gather2"],
         "bwd": ["loop", "transform+consts_dma", "barrierA", "gather1+consts_wait", "barrierB", "tile_dma_issue", "gather2", "tile_dma_wait", "-", "next_ids", "-"]}
def read(nblk, which):
    # [block][wave][16]: slots 0..11 = phase cycles of that wave, slots 12..15 of wave 0 = wall-clock marks
    buf = (C.c_ulonglong * (nblk * 256))()
    rc = getattr(lib, "gmc_debug_read_stamps_" + which)(buf, nblk * 256); assert rc == 0
    full = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 16, 16).astype(np.float64)
    read.waves = full                      # every wave's phase cycles
    return full[:, 0, :].copy()            # wave 0 (+ the marks), the shape the rest of this script expects
def all_waves(label, names):
    w = read.waves[:, :, :len(names)]      # [block][wave][phase]
    tot = w.sum(2, keepdims=True)
    share = 100 * w / tot
    print(label, "share of cycles per phase, mean over ALL waves:", {k: round(float(share[:, :, i].mean()), 1) for i, k in enumerate(names)})
    bar = [i for i, k in enumerate(names) if k.lower().startswith("barrier")]
    bs = share[:, :, bar].sum(2)
    print(label, "barrier share by wave (0 = oldest):", [round(float(bs[:, v].mean()), 1) for v in range(16)], "mean %.1f" % bs.mean())
for _ in range(3): eng.train_fwd_bwd(batch)
torch.cuda.synchronize()
# forward only -> stamps of fwd1 (grid 1280)
eng.forward(batch); torch.cuda.synchronize()
a = read(256, 'fwd')
tot = a[:, :12].sum(1).mean()
all_waves("fwd1", names["fwd"])
print("fwd1 cycles per WG by phase (wave 0):", {k: int(a[:, i].mean()) for i, k in enumerate(names["fwd"])}, "epilogue", int(a[:, 7].mean()), "prologue", int(a[:, 11].mean()))
print("fwd1 cycles per WG by phase (mean over waves):", {k: int(read.waves[:, :, i].mean()) for i, k in enumerate(names["fwd"])}, "prologue", int(read.waves[:, :, 11].mean()))
print("fwd1 cycles per WG %.0f" % tot, {k: round(100 * a[:, i].mean() / tot, 1) for i, k in enumerate(names["fwd"])}, "epilogue %.1f prologue %.1f" % (100 * a[:, 7].mean() / tot, 100 * a[:, 11].mean() / tot))
eng.train_fwd_bwd(batch); torch.cuda.synchronize()
b = read(256, 'bwd')
tot = b[:, :11].sum(1).mean()
all_waves("bwd1", names["bwd"])
print("bwd1 cycles per WG by phase (mean over waves):", {k: int(read.waves[:, :, i].mean()) for i, k in enumerate(names["bwd"])})
print("bwd1 cycles per WG %.0f" % tot, {k: round(100 * b[:, i].mean() / tot, 1) for i, k in enumerate(names["bwd"])})

def marks(x, label):
    m = x[:, 12:16] * 10.0 / 1e3          # 100 MHz ticks -> us
    t0 = m[:, 0].min()
    ent, l0, l1, ex = m[:, 0] - t0, m[:, 1] - t0, m[:, 2] - t0, m[:, 3] - t0
    q = lambda v: "min %.1f med %.1f max %.1f" % (v.min(), np.median(v), v.max())
    print(label, "us since the first workgroup's entry | entry:", q(ent), "| loop start:", q(l0), "| loop end:", q(l1) if l1.max() > 0 else "-", "| exit:", q(ex))
    print(label, "per workgroup: prologue", q(l0 - ent), "| loop", q((l1 if l1.max() > 0 else ex) - l0), "| epilogue", q(ex - l1) if l1.max() > 0 else "-")
marks(a, "fwd1")
marks(b, "bwd1")
mb = b[:, 12:16] * 10.0 / 1e3
loop = mb[:, 2] - mb[:, 1]
blk = np.arange(256)
print("bwd1 loop us by chunk (= XCD = block & 7):", [round(float(np.median(loop[(blk & 7) == c])), 1) for c in range(8)])
print("bwd1 loop us by slice (= block >> 3), first 8 / last 4:", [round(float(np.median(loop[(blk >> 3) == s])), 1) for s in list(range(8)) + [28, 29, 30, 31]])
print("bwd1 prologue us by chunk:", [round(float(np.median((mb[:, 1] - mb[:, 0])[(blk & 7) == c])), 1) for c in range(8)])
ma = a[:, 12:16] * 10.0 / 1e3
lf = ma[:, 3] - ma[:, 1]
print("fwd1 loop us by XCD (block & 7):", [round(float(np.median(lf[(blk & 7) == c])), 1) for c in range(8)], "by block/32:", [round(float(np.median(lf[(blk >> 5) == c])), 1) for c in range(8)])
