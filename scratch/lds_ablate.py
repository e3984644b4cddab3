import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json; sys.path.insert(0, %r)
import numpy as np, torch, networkx as nx
import gcn_max_cut_amd as pkg
B, n, d, F = 160, 1000, 7, 500
hs = [pkg.from_networkx(nx.random_regular_graph(d, n, seed=3000 + i)) for i in range(B)]
batch = pkg.GraphBatch(hs, None)
from gcn_max_cut_amd.Training import TrainingNeural as T
cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=F)
net, _, _ = T.setup_model_and_optimizer(cfg)
eng = net.engine()
for _ in range(3): eng.train_fwd_bwd(batch)
with pkg.hip.Probe(400) as p:
    for _ in range(10): eng.train_fwd_bwd(batch)
agg = {}
for k, v in p.records: agg.setdefault(k, []).append(v)
print(json.dumps({k: round(float(np.mean(v))*1e3, 1) for k, v in agg.items()}))
X = torch.randn(160000, 512, device="cuda"); Y = torch.empty_like(X)
for _ in range(3): Y.copy_(X)
torch.cuda.synchronize()
ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a,b in ev:
    a.record(); Y.copy_(X); b.record()
torch.cuda.synchronize()
print("copy us", np.median([a.elapsed_time(b) for a,b in ev])*1e3)
''' % ROOT
for dbg in sys.argv[1:] or ("0", "1", "2", "4", "3", "6", "5", "7"):
    env = dict(os.environ); env["GMC_LDS_DBG"] = dbg
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("dbg", dbg, out.stdout.strip() or out.stderr[-300:], flush=True)
