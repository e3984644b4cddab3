"""Config 5 of BASELINE.json (mixed sizes, inference + 200-iteration post-processing) through the
reference's harness surface: per-graph times the harness itself records, plus the batched decode."""
import sys, os, time, contextlib, io; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.Training import TrainingNeural as T
from gcn_max_cut_amd.Testing import TestingNeuralNetwork as TN
from gcn_max_cut_amd.DataGenerator import graphExtender as GE
sizes = [50, 100, 200, 300, 500]
graphs, terms = {}, {}
for i in range(50):
    n = sizes[i % 5]; graphs[i] = bench.regular_graph(n, 6 + i % 3, 9000 + i); terms[i] = bench.terminals_of(n, 9000 + i)
with contextlib.redirect_stdout(io.StringIO()):
    ds = GE.process_graphs_from_folder(graphs, terms, 1000)
cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=500)
torch.manual_seed(0); np.random.seed(0)
net, embed, opt = T.setup_model_and_optimizer(cfg)
with contextlib.redirect_stdout(io.StringIO()):
    TN.test_multiple_graphs(net, ds, sizes, 200, verbose=False)   # warm
    torch.cuda.synchronize(); t = time.perf_counter()
    res, by = TN.test_multiple_graphs(net, ds, sizes, 200, verbose=False)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t
print("test_multiple_graphs: %d graphs in %.1f ms = %.2f ms/graph (forward+argmax %.3f ms, 200-iter post-processing %.3f ms per graph, harness timers)"
      % (len(res), 1e3 * t1, 1e3 * t1 / len(res), 1e3 * np.mean([r['simple_time'] for r in res]), 1e3 * np.mean([r['post_time'] for r in res])))
with contextlib.redirect_stdout(io.StringIO()):
    TN.decode_dataset(net, ds, 200)
    torch.cuda.synchronize(); t = time.perf_counter()
    out = TN.decode_dataset(net, ds, 200)
    torch.cuda.synchronize(); t2 = time.perf_counter() - t
print("decode_dataset (one batched forward + one batched sampler launch): %d graphs in %.1f ms = %.3f ms/graph" % (len(out), 1e3 * t2, 1e3 * t2 / len(out)))
