"""Time line of ONE graph-step of the reference schedule (one Adam step per graph, hipGraph per epoch): the wall-clock
marks (s_memrealtime, one 100 MHz counter for the whole chip) the `make stamp NAME=marks DEFS=-DGMC_MARKS_ONLY` build
leaves behind in its four kernels, put on one axis.  The marks are those of the LAST launch of each kernel = the last
graph-step of the last replayed epoch.  usage: python scratch/seq_stamps.py [graphs] [lib-name]"""
import sys, os, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ["GCN_MAXCUT_LIB"] = os.path.join(root, "gcn-max-cut_amd", "lib", "libgcnmaxcut_%s.so" % (sys.argv[2] if len(sys.argv) > 2 else "marks"))
import ctypes as C, time, numpy as np, torch, networkx as nx
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.Training import TrainingNeural as T
sys.path.insert(0, os.path.join(root, "tests"))
from util import dataset_of

n, d, F = 1000, 7, 500
cfg = T.TrainingConfig(n_nodes=n, hidden_dim=F)
net, _, opt = T.setup_model_and_optimizer(cfg)
def regular(seed):
    g = nx.random_regular_graph(d, n, seed=seed)
    for u, v in g.edges():
        g[u][v]["weight"] = 1; g[u][v]["capacity"] = 1
    return g
graphs = {i: regular(3000 + i) for i in range(G)}
ds = dataset_of(graphs, {i: [3 + i % 5, 40 + i % 7, 500 + i % 11] for i in graphs})
assert len(ds) == G, len(ds)
tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=1, local_shard=True)
lib = pkg.hip.load()
for _ in range(4):
    tr.epoch(ds)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    tr.epoch(ds)
torch.cuda.synchronize()
us_step = 1e6 * (time.perf_counter() - t0) / (K * G)

def read(which, count):
    buf = (C.c_ulonglong * count)()
    rc = getattr(lib, "gmc_debug_read_stamps_" + which)(buf, count); assert rc == 0, rc
    return np.frombuffer(buf, dtype=np.uint64).astype(np.float64) * 10.0 / 1e3   # 100 MHz ticks -> us

fw = read("fwd", 256 * 256).reshape(256, 256)[:, 12:16]     # [block][entry, loop start, -, exit]
bw = read("bwd", 256 * 256).reshape(256, 256)[:, 12:16]     # [block][entry, loop start, loop end, exit]
hd = read("head", 32).reshape(2, 16)[:, :10]               # [oldest / youngest wave][mark]
head_in_bwd = hd.max() < fw[:, 0].max() - 1e3              # no head launch in this step: its marks are stale (or zero) ...
if head_in_bwd:                                            # ... the one-graph backward ran the head itself (block 0's marks)
    hd = read("bwdhead", 32).reshape(2, 16)[:, :10]
fn = read("fin", 2048 * 2).reshape(2048, 2)
# blocks of the last launch: the marks written latest (a 160-graph launch earlier in the process would have left older ones)
def live(x, col):
    t = x[:, col]; return x[t > t.max() - 200.0]
fw, bw, fn = live(fw, 0), live(bw, 0), live(fn, 0)
t_ref = fw[:, 0].min()
q = lambda v: "%.2f..%.2f" % (v.min() - t_ref, v.max() - t_ref)
print("us per graph-step (host clock, %d graphs x %d epochs): %.2f" % (G, K, us_step))
print("fwd1  %3d workgroups | entry %s | loop start %s | exit %s" % (len(fw), q(fw[:, 0]), q(fw[:, 1]), q(fw[:, 3])))
names = ["entry", "fold done", "barrier 1 passed", "phase 1 done", "barrier 2 passed", "phase 2 done", "block sum done",
         "loss stored", "phase 3 done", "stores acknowledged"]
for w, lab in ((0, "oldest wave  "), (1, "youngest wave")):
    print("head%s %s |" % (" (inside bwd1's workgroup 0)" if head_in_bwd else " ", lab), " | ".join("%s %.2f" % (names[i], hd[w, i] - t_ref) for i in range(10)))
print("bwd1  %3d workgroups | entry %s | loop start %s | loop end %s | exit %s" % (len(bw), q(bw[:, 0]), q(bw[:, 1]), q(bw[:, 2]), q(bw[:, 3])))
print("fin   %3d workgroups | entry %s | exit %s" % (len(fn), q(fn[:, 0]), q(fn[:, 1])))
out = {"us_per_graph_step_host": us_step, "graphs": G,
       "fwd1": {"workgroups": len(fw), "entry": [fw[:, 0].min() - t_ref, fw[:, 0].max() - t_ref], "loop_start": [fw[:, 1].min() - t_ref, fw[:, 1].max() - t_ref], "exit": [fw[:, 3].min() - t_ref, fw[:, 3].max() - t_ref]},
       "head": {"marks": names, "oldest_wave": list(hd[0] - t_ref), "youngest_wave": list(hd[1] - t_ref)},
       "bwd1": {"workgroups": len(bw), "entry": [bw[:, 0].min() - t_ref, bw[:, 0].max() - t_ref], "loop_start": [bw[:, 1].min() - t_ref, bw[:, 1].max() - t_ref], "loop_end": [bw[:, 2].min() - t_ref, bw[:, 2].max() - t_ref], "exit": [bw[:, 3].min() - t_ref, bw[:, 3].max() - t_ref]},
       "finish": {"workgroups": len(fn), "entry": [fn[:, 0].min() - t_ref, fn[:, 0].max() - t_ref], "exit": [fn[:, 1].min() - t_ref, fn[:, 1].max() - t_ref]}}
print(json.dumps(out))
