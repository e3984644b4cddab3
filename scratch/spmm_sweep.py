"""SpMM tuning sweep on the headline shape (R=160,000, d=7, F=500).  Each variant runs in a
fresh process (GMC_SPMM_TUNE is read once)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json; sys.path.insert(0, %r)
import numpy as np, torch, networkx as nx
import gcn_max_cut_amd as pkg
B, n, d, F = int(os.environ.get("SW_B", 160)), 1000, 7, 500
hs = [pkg.from_networkx(nx.random_regular_graph(d, n, seed=3000 + i)) for i in range(B)]
batch = pkg.GraphBatch(hs, None)
lib = pkg.hip.load(); p = pkg.hip.ptr
X = torch.randn(batch.R, F, device="cuda"); Y = torch.empty_like(X)
bias = torch.randn(F, device="cuda"); W2 = torch.randn(F, 3, device="cuda"); Z0 = torch.empty(batch.R, 3, device="cuda")
def run(epi, group):
    return lib.gmc_spmm_f32(p(batch.rowptr), p(batch.gcol), None, p(batch.dinv), p(X), F, p(bias), 1, p(Y), F,
                            batch.R, F, group, p(W2) if epi else None, p(Z0) if epi else None, pkg.hip.stream())
res = {}
for epi in (0, 1):
    for group in (0, n):
        for _ in range(3): run(epi, group)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); rc = run(epi, group); b.record()
        torch.cuda.synchronize()
        t = np.median([a.elapsed_time(b) for a, b in ev]) * 1e-3
        res["epi%%d_group%%d" %% (epi, group)] = round(batch.spmm_bytes(F) / t / 1e9, 1)
print(json.dumps(res))
''' % ROOT
variants = [""] + [f"{r},{u},{n}" for n in (0, 1) for r in (1, 2, 4) for u in (4, 8)]
for v in variants:
    env = dict(os.environ)
    if v: env["GMC_SPMM_TUNE"] = v
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print((v or "default").ljust(8), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
