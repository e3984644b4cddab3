"""Where the non-kernel time of one batched step goes (host side), one GPU."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, networkx as nx
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.Training import TrainingNeural as T
from gcn_max_cut_amd.DataGenerator import graphExtender as GE
import contextlib, io
B, n, d = 160, 1000, 7
import bench
graphs = {i: bench.regular_graph(n, d, 3000 + i) for i in range(B)}
terms = {i: bench.terminals_of(n, 3000 + i) for i in range(B)}
with contextlib.redirect_stdout(io.StringIO()):
    ds = GE.process_graphs_from_folder(graphs, terms, 1000)
cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=500)
net, embed, opt = T.setup_model_and_optimizer(cfg)
tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=B)
for _ in range(5): tr.epoch(ds)
torch.cuda.synchronize()
def timeit(f, k=200):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e6
print("epoch()                     %.1f us" % timeit(lambda: tr.epoch(ds)))
print('use_graph', tr._use_graph(), tr._graph, tr.allow_graph, tr.world, len(tr._batches))
g = tr._graph
print("replay + stream sync        %.1f us" % timeit(lambda: (g.replay(), torch.cuda.current_stream().synchronize())))
print("replay only (back to back)  %.1f us" % timeit(lambda: g.replay()))
ev = torch.cuda.Event()
def with_event():
    g.replay(); ev.record(); ev.synchronize()
print("replay + event sync         %.1f us" % timeit(with_event))
t = time.perf_counter()
for _ in range(200): tr.prepare(ds)
print("prepare() (cached)          %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
host = tr._loss_host.numpy()
t = time.perf_counter()
for _ in range(200):
    tot = 0.0
    for i, batch in enumerate(tr._batches): tot += float(host[i, :batch.B].sum(dtype=np.float32))
print("host loss sum               %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
