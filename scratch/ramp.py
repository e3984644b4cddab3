"""Per-step wall time of the first replays after start-up (is the slow start the graph upload or the clocks?)."""
import sys, os, time, contextlib, io; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.Training import TrainingNeural as T
from gcn_max_cut_amd.DataGenerator import graphExtender as GE
B, n, d = 160, 1000, 7
graphs = {i: bench.regular_graph(n, d, 3000 + i) for i in range(B)}
terms = {i: bench.terminals_of(n, 3000 + i) for i in range(B)}
with contextlib.redirect_stdout(io.StringIO()):
    ds = GE.process_graphs_from_folder(graphs, terms, 1000)
cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=500)
net, embed, opt = T.setup_model_and_optimizer(cfg)
tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=B)
tr.prepare(ds)
time.sleep(3.0)   # idle GPU, like after the dataset build
ts = []
for i in range(80):
    torch.cuda.synchronize(); t = time.perf_counter(); tr.epoch(ds); ts.append((time.perf_counter() - t) * 1e3)
print("ms per step:", " ".join("%.3f" % x for x in ts[:16]), "... steps 20-29 mean %.3f, 40-49 %.3f, 70-79 %.3f" % (np.mean(ts[20:30]), np.mean(ts[40:50]), np.mean(ts[70:80])))
