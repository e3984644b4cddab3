#!/usr/bin/env python3
"""Generates tests/golden/training.json - run in the build container only (needs /root/reference).

Pins every part of the hot path that the reference's OWN code can pin.
``python/Training/TrainingNeural.py`` and ``python/DataGenerator/graphExtender.py`` star-import
``python/commons.py``, which imports ``dgl`` (dgl==2.0.0, ``envList.txt:36``; absent here).  This
script puts an INERT placeholder named ``dgl`` on ``sys.modules`` so that those imports resolve:

* ``dgl.nn.pytorch.GraphConv`` RAISES when constructed - no GraphConv arithmetic is stood in for;
* ``dgl.from_networkx`` returns an opaque token (``.to()`` returns itself) - no graph conversion.

Everything recorded below is therefore the output of the reference's own pure-torch / networkx code:
``override_fixed_nodes`` (:87-94), ``apply_max_to_one_hot`` (:96-106), ``extend_matrix_torch`` (:137-152),
``calculate_HC_vectorized`` / ``compute_loss`` (:154-176, :291-309) and autograd ``dL/dP`` of that chain,
``train_model`` (:392-484; ``train_single_epoch`` and ``setup_model_and_optimizer`` patched to scripted
stand-ins, ``torch.save`` patched to a recorder), ``evaluate_model`` (:537-570; a model object that returns
stored probabilities), ``evaluate_optimal_partitioning`` / ``calculate_all_cut_legacy`` (:229-289),
``save_neural_model`` key set (:611-634), ``graphExtender.swap_graph_nodes`` (:8-26), ``extend_matrix_torch_2``
(:28-48), ``process_graphs_from_folder`` (:50-132: four terminal cases, the skip, the batch flush) and
``commons.gen_adj_matrix`` / ``qubo_dict_to_torch`` (:38-77).

What stays UNPINNED (the environment's limit): ``GraphConv.forward`` and ``dgl.from_networkx``.
Only data (inputs / expected outputs) is written; no reference source text, no pickled reference classes.
"""
import contextlib
import io
import json
import os
import re
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


class _GraphToken:
    """What the placeholder ``dgl.from_networkx`` hands back: opaque, carries nothing."""

    def to(self, *_a, **_k):
        return self


def _install_inert_dgl():
    class GraphConv:  # noqa: D401 - inert: constructing it is an error, there is no forward
        def __init__(self, *a, **k):
            raise RuntimeError("inert dgl placeholder: GraphConv has no implementation here")

    dgl = types.ModuleType("dgl")
    dgl.__inert_placeholder__ = True
    dgl.from_networkx = lambda nx_graph=None, **_k: _GraphToken()
    nn_mod = types.ModuleType("dgl.nn")
    pt_mod = types.ModuleType("dgl.nn.pytorch")
    pt_mod.GraphConv = GraphConv
    nn_mod.pytorch = pt_mod
    dgl.nn = nn_mod
    sys.modules.update({"dgl": dgl, "dgl.nn": nn_mod, "dgl.nn.pytorch": pt_mod})


_install_inert_dgl()
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
import networkx as nx  # noqa: E402
import torch  # noqa: E402
from python.DataGenerator.GraphCreator import generate_graph  # noqa: E402
from python.Training import TrainingNeural as RT  # noqa: E402
from python.DataGenerator import graphExtender as RG  # noqa: E402
from python import commons as RC  # noqa: E402


def f32hex(x) -> str:
    return np.float32(x).tobytes().hex()


def triplets(t: torch.Tensor):
    """Non-zeros of a 2-D tensor as [row, col, value] - pins the dense matrix exactly."""
    nz = torch.nonzero(t)
    return [[int(r), int(c), float(t[r, c])] for r, c in nz.tolist()]


def edges_of(g):
    return [[int(u), int(v), float(d.get("weight", 1))] for u, v, d in g.edges(data=True)]


def weighted(g, seed):
    """integer weights 1..5 on the edges (the loss multiplies by them; GraphConv would not)."""
    rng = np.random.RandomState(seed)
    for u, v in g.edges():
        g[u][v]["weight"] = int(rng.randint(1, 6))
    return g


def probs_for(n, seed):
    rng = np.random.RandomState(seed)
    z = rng.standard_normal((n, 3)).astype(np.float32) * 2.0
    e = np.exp(z - z.max(1, keepdims=True))
    p = (e / e.sum(1, keepdims=True)).astype(np.float32)
    p[5] = [0.5, 0.5, 0.0]        # exact tie: first maximum wins (:98)
    p[6] = [0.25, 0.375, 0.375]   # exact tie between classes 1 and 2
    p[7] = [0.0, 0.0, 1.0]        # saturated row (neural_network_testing.ipynb:1060-1063)
    p[8] = [1.0 / 3, 1.0 / 3, 1.0 / 3]  # three-way tie in fp32
    p[1] = [0.9, 0.05, 0.05]      # a terminal whose argmax disagrees with its forced class
    return p


def loss_cases():
    out = []
    for (n, d, s, wseed, Cc) in [(50, 6, 1001, None, 1.0), (100, 7, 1000, None, 1.0), (50, 6, 1001, 7, 1.0),
                                 (100, 7, 1000, 11, 2.5), (64, 8, 1002, None, 0.5)]:
        g = generate_graph(n=n, d=d, graph_type="reg", random_seed=s)
        if wseed is not None:
            weighted(g, wseed)
        a_pad = RG.extend_matrix_torch_2(RC.qubo_dict_to_torch(g, RC.gen_adj_matrix(g), torch_dtype=torch.float32), 1000,
                                         torch_dtype=torch.float32)
        p_np = probs_for(n, s)
        h = torch.from_numpy(p_np.copy()).requires_grad_(True)
        ov = RT.override_fixed_nodes(h)
        hot = RT.apply_max_to_one_hot(ov)
        loss = RT.compute_loss(hot, a_pad, 0.0, Cc, 1000.0)
        loss.backward()
        hc = RT.calculate_HC_vectorized(hot.detach(), a_pad)
        out.append({
            "n": n, "d": d, "seed": s, "weight_seed": wseed, "C": Cc,
            "edges": edges_of(g),
            "adjacency_shape": list(a_pad.shape), "adjacency_nonzeros": triplets(a_pad),
            "P": p_np.tolist(),
            "override": ov.detach().numpy().tolist(),
            "override_hex": [[f32hex(v) for v in row] for row in ov.detach().numpy()[:3]],
            "one_hot": hot.detach().numpy().tolist(),
            "argmax": [int(i) for i in torch.argmax(hot.detach(), dim=1)],
            "loss": float(loss.item()), "loss_hex": f32hex(loss.item()),
            "HC": float(hc.item()),
            "dL_dP": h.grad.numpy().tolist(),
        })
    return out


def extend_cases():
    m = torch.arange(12, dtype=torch.float32).reshape(3, 4)[:, :3] + 1
    a = RT.extend_matrix_torch(m, 7)
    b = RG.extend_matrix_torch_2(m, 7)
    c = RT.extend_matrix_torch(m, 3)
    errs = {}
    for name, fn in (("extend_matrix_torch", RT.extend_matrix_torch), ("extend_matrix_torch_2", RG.extend_matrix_torch_2)):
        try:
            fn(m, 2)
            errs[name] = None
        except ValueError as e:
            errs[name] = str(e)
    return {"input": m.tolist(), "N": 7, "extend_matrix_torch": a.tolist(), "extend_matrix_torch_2": b.tolist(),
            "extend_same_size": c.tolist(), "dtype": str(a.dtype), "too_small_error": errs}


def swap_cases():
    out = []
    for (n, d, s, mapping) in [(12, 3, 5, {7: 0, 9: 1, 11: 2, 0: 7, 1: 9, 2: 11}), (10, 3, 6, {4: 0, 8: 1, 0: 4, 1: 8}),
                               (10, 3, 7, {5: 1, 1: 5})]:
        g = weighted(generate_graph(n=n, d=d, graph_type="reg", random_seed=s), s)
        before = edges_of(g)
        RG.swap_graph_nodes(g, dict(mapping))
        out.append({"n": n, "d": d, "seed": s, "mapping": [[int(k), int(v)] for k, v in mapping.items()],
                    "edges_before": before, "edges_after": edges_of(g), "nodes_after": [int(u) for u in g.nodes()]})
    return out


def extender_cases():
    """process_graphs_from_folder: the four terminal cases, the skip branch, the batch flush."""
    specs = [  # (n, d, seed, terminals)
        (20, 3, 21, [7, 13, 4]),     # none of 0,1,2 is a terminal
        (20, 3, 22, [11, 2, 6]),     # only 2
        (24, 5, 23, [9, 1, 17]),     # only 1
        (24, 5, 24, [0, 19, 8]),     # only 0
        (20, 3, 25, [0, 1, 9]),      # two of them: skipped
        (30, 4, 26, [29, 3, 15]),    # none again (after a skip: index continues from the kept count)
    ]
    graphs = {f"g{i}": weighted(generate_graph(n=n, d=d, graph_type="reg", random_seed=s), s) for i, (n, d, s, _t) in enumerate(specs)}
    terms = {f"g{i}": list(t) for i, (_n, _d, _s, t) in enumerate(specs)}
    inputs = [{"name": k, "n": specs[i][0], "d": specs[i][1], "seed": specs[i][2], "terminals": list(specs[i][3]),
               "edges": edges_of(g)} for i, (k, g) in enumerate(graphs.items())]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ds = RG.process_graphs_from_folder(graphs, terms, 1000)
    items = []
    for key, (tok, a_pad, g, t) in ds.items():
        assert isinstance(tok, _GraphToken)
        items.append({"key": int(key), "edges_after": edges_of(g), "shape": list(a_pad.shape), "dtype": str(a_pad.dtype),
                      "adjacency_nonzeros": triplets(a_pad), "terminals_entry": list(t)})
    res = {"inputs": inputs, "max_nodes": 1000, "items": items,
           "terminals_after": {k: list(v) for k, v in terms.items()},
           "stdout": buf.getvalue().splitlines()}

    # batch flush (Q11): dict reset whenever i % save_batch_size == 0
    saved = []
    orig = RG.save_object
    RG.save_object = lambda obj, filename: saved.append({"filename": filename, "keys": [int(k) for k in obj.keys()]})
    try:
        flush = {}
        for bs in (2, 5):
            graphs2 = {f"g{i}": generate_graph(n=n, d=d, graph_type="reg", random_seed=s) for i, (n, d, s, _t) in enumerate(specs)}
            terms2 = {f"g{i}": list(t) for i, (_n, _d, _s, t) in enumerate(specs)}
            saved.clear()
            with contextlib.redirect_stdout(io.StringIO()):
                ds2 = RG.process_graphs_from_folder(graphs2, terms2, 1000, save_batch_size=bs, output_filename_prefix="pfx")
            flush[str(bs)] = {"saved": [dict(x) for x in saved], "returned_keys": [int(k) for k in ds2.keys()]}
    finally:
        RG.save_object = orig
    res["batch_flush"] = flush

    # max_nodes smaller than the graph: the reference swallows the ValueError and returns what it has
    graphs3 = {"a": generate_graph(n=20, d=3, graph_type="reg", random_seed=21)}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ds3 = RG.process_graphs_from_folder(graphs3, {"a": [7, 13, 4]}, 10)
    res["too_small"] = {"returned_keys": [int(k) for k in ds3.keys()],
                        "mentions_value_error": "N should be greater than or equal to the original matrix size." in buf.getvalue()}
    return res


def adjacency_cases():
    out = []
    for (n, d, s) in [(16, 3, 31), (30, 5, 32)]:
        g = weighted(generate_graph(n=n, d=d, graph_type="reg", random_seed=s), s)
        q = RC.gen_adj_matrix(g)
        m = RC.qubo_dict_to_torch(g, q, torch_dtype=torch.float32)
        out.append({"n": n, "d": d, "seed": s, "edges": edges_of(g), "dict_entries": len(q), "shape": list(m.shape),
                    "dtype": str(m.dtype), "nonzeros": triplets(m)})
    return out


class _ScriptedNet(torch.nn.Module):
    """Stand-in for the DGL model inside train_model's bookkeeping (ours, not the reference's)."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(2))


def train_model_cases():
    scripts = {
        "rise_then_stop": dict(losses=[-10.0, -12.0, -11.0, -11.5, -11.2, -11.1, -13.0, -12.5], tolerance=1e-4, patience=3,
                               save_frequency=2, number_epochs=8, save_directory="m.pth"),
        "plateau": dict(losses=[-5.0, -5.00005, -5.00007, -5.00008, -7.0], tolerance=1e-4, patience=3, save_frequency=100,
                        number_epochs=5, save_directory="m.pth"),
        "reset_counter": dict(losses=[-1.0, -0.5, -2.0, -1.5, -3.0, -2.5, -4.0], tolerance=1e-4, patience=2, save_frequency=3,
                              number_epochs=7, save_directory=None),
        "runs_out": dict(losses=[-1.0, -2.0, -3.0], tolerance=1e-4, patience=20, save_frequency=1, number_epochs=3,
                         save_directory="x.pth"),
    }
    out = {}
    for name, sc in scripts.items():
        it = iter(sc["losses"])
        saved = []
        orig = (RT.train_single_epoch, RT.setup_model_and_optimizer, torch.save)
        net = _ScriptedNet()
        embed = torch.nn.Embedding(4, 4)
        opt = torch.optim.Adam(list(net.parameters()) + list(embed.parameters()), lr=1e-3)
        RT.train_single_epoch = lambda *a, **k: next(it)
        RT.setup_model_and_optimizer = lambda cfg: (net, embed, opt)
        torch.save = lambda obj, f, *a, **k: saved.append({"filename": f, "keys": list(obj.keys()), "epoch": int(obj["epoch"]),
                                                           "loss_history": [float(x) for x in obj["loss_history"]]})
        cfg = RT.TrainingConfig(n_nodes=1000, number_epochs=sc["number_epochs"], tolerance=sc["tolerance"], patience=sc["patience"],
                                save_frequency=sc["save_frequency"], save_directory=sc["save_directory"])
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                r_net, best, epoch, emb_w, hist = RT.train_model({}, cfg)
        finally:
            RT.train_single_epoch, RT.setup_model_and_optimizer, torch.save = orig
        lines = [re.sub(r"Training completed in [0-9.]+ seconds", "Training completed in <T> seconds", ln)
                 for ln in buf.getvalue().splitlines()]
        lines = [re.sub(r"^Device: .*", "Device: <D>", ln) for ln in lines]
        out[name] = {"config": {k: sc[k] for k in ("tolerance", "patience", "save_frequency", "number_epochs", "save_directory")},
                     "losses": sc["losses"], "best_loss": float(best), "final_epoch": int(epoch),
                     "loss_history": [float(x) for x in hist], "returns_same_net": r_net is net,
                     "returns_embed_weight": emb_w is embed.weight, "saved": saved, "stdout": lines}
    return out


def evaluate_cases():
    out = []
    for specs in ([(50, 6, 1001), (64, 8, 1002)], []):
        ds, stored = {}, []
        for i, (n, d, s) in enumerate(specs):
            g = generate_graph(n=n, d=d, graph_type="reg", random_seed=s)
            a_pad = RG.extend_matrix_torch_2(RC.qubo_dict_to_torch(g, RC.gen_adj_matrix(g), torch_dtype=torch.float32), 1000)
            p = probs_for(n, s + 17)
            ds[i] = [("tok", i), a_pad, g, [0, 1, 2]]
            stored.append({"n": n, "d": d, "seed": s, "edges": edges_of(g), "P": p.tolist()})

        class Model:
            def eval(self):
                return self

            def __call__(self, tok, a_pad):
                return torch.tensor(stored[tok[1]]["P"], dtype=torch.float32)

        cfg = RT.TrainingConfig(n_nodes=1000, C=1.0)
        res = RT.evaluate_model(Model(), ds, cfg)
        out.append({"graphs": stored, "result": {k: (float(v) if k != "num_samples" else int(v)) for k, v in res.items()}})
    return out


def optimal_partitioning_cases():
    out = []
    for (n, d, s) in [(40, 5, 41), (20, 3, 42)]:
        g = weighted(generate_graph(n=n, d=d, graph_type="reg", random_seed=s), s)
        adj = RC.qubo_dict_to_torch(g, RC.gen_adj_matrix(g), torch_dtype=torch.float32)
        p = probs_for(n, s)
        seen_inputs = []

        class Tok:
            def number_of_nodes(self):
                return n

        class Net:
            def eval(self):
                return self

            def __call__(self, tok, inputs):
                seen_inputs.append(list(inputs.shape))
                return torch.from_numpy(p.copy())

        best = RT.evaluate_optimal_partitioning(Net(), Tok(), torch.zeros(n, 1000), adj, {"s": 0, "t": 1, "u": 2})
        binary = (RT.override_fixed_nodes(torch.from_numpy(p.copy())) >= 0.5).float()
        out.append({"n": n, "d": d, "seed": s, "edges": edges_of(g), "P": p.tolist(), "best_loss": float(best),
                    "model_calls": len(seen_inputs), "input_shapes": seen_inputs[:1],
                    "binary": binary.numpy().tolist(),
                    "calculate_all_cut_legacy": float(RT.calculate_all_cut_legacy(adj, binary)),
                    "permutations": [{k: int(v) for k, v in dct.items()} for dct in RT.generate_terminal_permutations({"s": 0, "t": 1, "u": 2})]})
    return out


def save_case():
    net = _ScriptedNet()
    embed = torch.nn.Embedding(4, 4)
    opt = torch.optim.Adam(list(net.parameters()) + list(embed.parameters()), lr=1e-3)
    got = {}
    orig = torch.save
    torch.save = lambda obj, f, *a, **k: got.update(filename=f, keys=list(obj.keys()), inputs_is_embed_weight=obj["inputs"] is embed.weight,
                                                    epoch=obj["epoch"], loss_history=list(obj["loss_history"]))
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            RT.save_neural_model(net, opt, embed, 7, [-1.0, -2.0], RT.TrainingConfig(), "some/path.pth")
    finally:
        torch.save = orig
    got["stdout"] = buf.getvalue().splitlines()
    return got


def config_case():
    c = RT.TrainingConfig()
    d = RT.TrainingConfig(n_nodes=300, hidden_dim=7)
    return {"defaults": dict(c.__dict__), "derived": dict(d.__dict__), "fields": list(RT.TrainingConfig.__dataclass_fields__)}


def main():
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)  # nothing the reference writes may land in the repo
        try:
            data = {
                "source": "reference python/Training/TrainingNeural.py, python/DataGenerator/graphExtender.py, python/commons.py "
                          "imported from /root/reference and run in the build container",
                "dgl": "INERT placeholder on sys.modules: GraphConv raises on construction, from_networkx returns an opaque token; "
                       "no arithmetic stand-in. GraphConv.forward and dgl.from_networkx remain parity-unpinned.",
                "versions": {"torch": torch.__version__, "numpy": np.__version__, "networkx": nx.__version__},
                "config": config_case(),
                "loss": loss_cases(),
                "extend": extend_cases(),
                "swap": swap_cases(),
                "extender": extender_cases(),
                "adjacency": adjacency_cases(),
                "train_model": train_model_cases(),
                "evaluate_model": evaluate_cases(),
                "optimal_partitioning": optimal_partitioning_cases(),
                "save_neural_model": save_case(),
            }
        finally:
            os.chdir(cwd)
    path = os.path.join(HERE, "training.json")
    json.dump(data, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
