"""Where does a replayed step's host time go?  Per step: replay() call duration, time until the losses are
visible in the pinned buffer, time until the stream is idle."""
import os, sys, time, io, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.DataGenerator import graphExtender as GE
from gcn_max_cut_amd.Training import TrainingNeural as T
gpg = 160
graphs = {i: bench.regular_graph(1000, 7, 3000 + i) for i in range(gpg)}
terms = {i: bench.terminals_of(1000, 3000 + i) for i in range(gpg)}
with contextlib.redirect_stdout(io.StringIO()):
    ds = GE.process_graphs_from_folder(graphs, terms, 1000)
cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=500, number_epochs=10**9, patience=10**9)
torch.manual_seed(0)
net, embed, opt = T.setup_model_and_optimizer(cfg)
tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=gpg, local_shard=True)
net.train()
for _ in range(30):
    tr.epoch(ds)
torch.cuda.synchronize()
host = tr._loss_host_np
rec = []
for it in range(200):
    host.fill(np.nan)
    t0 = time.perf_counter()
    tr._graph.replay()
    t1 = time.perf_counter()
    while host[0, 159] != host[0, 159] or host[0, 0] != host[0, 0]:
        pass
    t2 = time.perf_counter()
    torch.cuda.current_stream().synchronize()
    t3 = time.perf_counter()
    rec.append((t1 - t0, t2 - t0, t3 - t0))
a = np.array(rec[20:]) * 1e6
print("zero-copy" if tr._loss_host_dev else "copy node", "replay() call %.1f us | losses visible %.1f us | stream idle %.1f us (medians)" % tuple(np.median(a, axis=0)))
# back-to-back replays without waiting: does a second launch queue behind the first?
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    tr._graph.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("100 replays back to back: enqueue %.1f us each, total %.1f us each" % ((t1 - t0) * 1e4, (t2 - t0) * 1e4))
