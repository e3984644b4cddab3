"""The CPU oracle against everything the reference lets us pin:
golden vectors produced by the reference's own importable modules
(tests/golden/make_golden.py), an independent float64 formulation of GraphConv, the
second oracle implementation, and the invariants the reference's notebooks print."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R
from tests import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def test_graph_generator_matches_reference_generate_graph():
    """oracle.regular_graph / bench.regular_graph == GraphCreator.generate_graph (golden)."""
    import bench
    for rec in load("graphs.json")["graphs"]:
        for make in (R.regular_graph, bench.regular_graph):
            g = make(rec["n"], rec["d"], rec["seed"])
            assert list(g.nodes()) == rec["nodes"]
            got = [[int(u), int(v), int(w["weight"]), int(w["capacity"])] for u, v, w in g.edges(data=True)]
            assert got == rec["edges"]


def test_decode_cut_and_sampler_match_reference():
    for case in load("decode.json")["cases"]:
        g = R.regular_graph(case["n"], case["d"], case["seed"])
        probs = np.asarray(case["probs"], np.float32)
        part = R.partition_of(torch.from_numpy(probs)).tolist()
        assert part == case["simple_assignment"]          # first max on ties, terminals forced
        assert R.cut_value(part, g) == case["simple_cut"]
        np.random.seed(case["np_seed"])
        assert R.sample_partition(probs, np.random.rand) == case["sample0"]
        assert R.sample_partition(probs, np.random.rand) == case["sample1"]
        assert R.cut_value(case["sample0"], g) == case["sample0_cut"]
        np.random.seed(case["np_seed"])
        best, cut = R.post_process(probs, g, case["post_iterations"], np.random.rand)
        assert best == case["post_assignment"] and cut == case["post_cut"]
        # the C oracle's cut agrees too
        rp, cl, vl = CO.csr_of(g)
        S, loss, _ = CO.loss_grad(rp, cl, vl, probs)
        assert S.tolist() == part and loss == -float(case["simple_cut"])


SPECS = [(100, 7, 1000), (50, 6, 1001), (64, 8, 1002), (30, 5, 1003)]


def test_loss_is_minus_cut_of_reference_decode():
    """loss == -calculate_cut_value(simple_partition_assignment(P)) - integer valued."""
    ds = R.synthetic_dataset(SPECS)
    params = R.init_params(1000, 16, 3, seed=3)
    for (g, a_pad, nx_g, _t) in ds.values():
        loss, P = R.graph_loss(params, g, a_pad)
        cut = R.cut_value(R.partition_of(P).tolist(), nx_g)
        assert abs(float(loss) + cut) < 1e-3 and float(loss) == pytest.approx(round(float(loss)), abs=1e-3)
        assert g.number_of_edges() == 2 * nx_g.number_of_edges()     # ...pipeline.ipynb:442
        assert tuple(a_pad.shape) == (nx_g.number_of_nodes(), 1000)   # ...pipeline.ipynb:443


def test_parameter_count_matches_notebooks():
    assert sum(v.numel() for v in R.init_params().values()) == 502003  # ...pipeline.ipynb:698


def test_graphconv_restatement_against_dense_float64_formula():
    ds = R.synthetic_dataset(SPECS)
    params = R.init_params(1000, 32, 3, seed=1)
    for (g, a_pad, nx_g, _t) in ds.values():
        P = R.forward(params, g, a_pad).numpy()
        assert np.abs(P - R.forward_dense_f64(params, nx_g)).max() < 5e-7


def test_two_oracles_agree_on_forward_gradients_and_training():
    ds = R.synthetic_dataset(SPECS)
    params = R.init_params(1000, 16, 3, seed=0)
    pn = {k: v.numpy().copy() for k, v in params.items()}
    items = list(ds.values())
    loss, grads, probs = R.loss_and_grads(params, items)
    ct = CO.CTrainer(pn)
    closs = ct.step([CO.csr_of(it[2]) for it in items])
    assert float(closs.sum()) == loss
    o = np.cumsum([0, 1000 * 16, 16, 48, 3])
    for i, k in enumerate(("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias")):
        r = grads[k].numpy().ravel()
        assert np.abs(ct.grad[o[i]:o[i + 1]] - r).max() <= 1e-5 * max(1.0, np.abs(r).max())
    # dW1 rows no graph reaches are exactly zero
    assert np.abs(ct.grad[100 * 16:1000 * 16]).max() == 0.0
    # three reference-schedule epochs (one Adam step per graph)
    tr, ct2 = R.Trainer(params), CO.CTrainer(pn)
    csrs = [CO.csr_of(it[2]) for it in items]
    for _ in range(3):
        assert tr.epoch_sequential(ds) == float(sum(float(ct2.step([c])[0]) for c in csrs))
    for k, v in tr.state().items():
        assert np.abs(v.numpy() - ct2.unpack()[k]).max() < 1e-6


def test_c_adam_matches_torch_adam():
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(4099, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.numpy().copy(), np.zeros(4099, np.float32), np.zeros(4099, np.float32)
    for step in range(1, 5):
        grad = torch.randn(4099, generator=g)
        ref.grad = grad.clone()
        opt.step()
        CO.adam(p, grad.numpy().copy(), m, v, 1e-3, step)
        np.testing.assert_allclose(p, ref.detach().numpy(), rtol=3e-7, atol=3e-7)


def test_early_stop_trace_semantics():
    # rises or stalls (<= tol) count; the break comes after the append (Q8)
    stop, best, hist = R.early_stop_trace([-10, -12, -12, -11, -13, -13.00001, -13.00001], 1e-4, 3)
    assert stop is None and best == -13.00001 and len(hist) == 7
    stop, best, hist = R.early_stop_trace([-10, -9, -8, -7, -20], 1e-4, 3)
    assert stop == 3 and best == -10 and hist == [-10, -9, -8, -7]
    stop, _, _ = R.early_stop_trace([-5, -5, -5], 1e-4, 2)
    assert stop == 2


def test_terminal_normalisation_cases_and_skip():
    import networkx as nx
    base = R.regular_graph(20, 3, 1)
    for terms, kept in ([5, 9, 11], True), ([2, 7, 9], True), ([1, 5, 6], True), ([0, 7, 8], True), ([0, 1, 5], False), ([1, 2, 9], False):
        g = base.copy()
        before = sorted(sorted(e) for e in g.edges())
        t = list(terms)
        ok = R.normalise_terminals(g, t)
        assert ok == kept
        if ok:
            assert sorted(g.nodes()) == list(range(20)) and g.number_of_edges() == len(before)
            assert nx.is_isomorphic(g, base)
