"""Where does the G(n,p) step's time go?  Fused step timing of: d=16 regular (16 slots, no overflow), G(n,p) as is,
G(n,p) with every node's degree capped at 16 (edges dropped: timing only), d=7 + one hub per graph."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, networkx as nx, torch
import bench
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.DataGenerator import graphExtender as GE
from gcn_max_cut_amd.Training import TrainingNeural as T

B, n = int(os.environ.get("B", "160")), 1000
def capped(g, cap=16):
    g = g.copy()
    for v in list(g.nodes()):
        while g.degree(v) > cap:
            u = max(g.neighbors(v), key=g.degree)
            g.remove_edge(v, u)
    for v in g.nodes():
        if g.degree(v) == 0:
            g.add_edge(v, (v + 1) % n, weight=1, capacity=1)
    return g
def hub(g, deg=40):
    g = g.copy()
    for v in range(10, 10 + deg - 7):
        g.add_edge(5, v, weight=1, capacity=1)
    return g
sets = {"d16": [bench.regular_graph(n, 16, 100 + i) for i in range(B)],
        "gnp": [bench.gnp_graph(n, 0.01, 200 + i) for i in range(B)]}
sets["gnp_capped16"] = [capped(g) for g in sets["gnp"]]
sets["d7_hub40_every_graph"] = [hub(bench.regular_graph(n, 7, 300 + i)) for i in range(B)]
sets["d7_hub40_one_graph"] = [bench.regular_graph(n, 7, 300 + i) for i in range(B - 1)] + [hub(bench.regular_graph(n, 7, 999))]
cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=500)
for name, gl in sets.items():
    gs = dict(enumerate(gl)); ts = {i: bench.terminals_of(n, 7 + i) for i in gs}
    with contextlib.redirect_stdout(io.StringIO()):
        ds = GE.process_graphs_from_folder(gs, ts, 1000)
    torch.manual_seed(0)
    net, embed, opt = T.setup_model_and_optimizer(cfg); net.train()
    tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=len(ds), local_shard=True)
    tr.allow_graph = False
    for _ in range(20): tr.epoch(ds)
    torch.cuda.synchronize()
    with pkg.hip.Probe(16 * 30) as p:
        for _ in range(30): tr.epoch(ds)
        torch.cuda.synchronize()
    by = {}
    for t, ms in p.records: by.setdefault(t, []).append(ms * 1e3)
    h = tr._batches[0].host
    print(name, "W", h.ell_width, "slots", h.ell_slots, "maxdeg", h.max_degree, "ovf blocks", 0 if h.ovf_ptr is None else int(h.ovf_ptr[-1]),
          {k: round(float(np.mean(v)), 1) for k, v in by.items()}, flush=True)
