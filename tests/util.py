"""Shared builders for the parity tests: the same seeded graphs go through the product's
GraphExtender (-> GraphHandle + padded adjacency) and through the oracle's."""
import copy

import numpy as np
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R


def make_graphs(specs):
    graphs = {i: R.regular_graph(n, d, s) for i, (n, d, s) in enumerate(specs)}
    terms = {i: R.seeded_terminals(n, s) for i, (n, d, s) in enumerate(specs)}
    return graphs, terms


def product_dataset(specs, max_nodes=1000):
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    graphs, terms = make_graphs(specs)
    return GE.process_graphs_from_folder(graphs, terms, max_nodes)


def oracle_dataset(specs, max_nodes=1000):
    graphs, terms = make_graphs(specs)
    return R.make_dataset(graphs, terms, max_nodes)


def csrs_of(dataset):
    return [CO.csr_of(item[2]) for item in dataset.values()]


def np_params(state):
    return {k: v.detach().cpu().numpy().astype(np.float32).copy() for k, v in state.items()}


def weighted_copy(dataset_specs, seed=0):
    """Graphs with non-unit integer weights (exercises the `vals` path)."""
    graphs, terms = make_graphs(dataset_specs)
    rng = np.random.RandomState(seed)
    for g in graphs.values():
        for u, v in g.edges():
            g[u][v]["weight"] = int(rng.randint(1, 4))
    return graphs, terms


def dataset_of(graphs, terms, max_nodes=1000):
    """Product dataset of arbitrary networkx graphs (dict index -> graph / terminal list)."""
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    return GE.process_graphs_from_folder(graphs, terms, max_nodes)


def check_step_against_oracle(pkg, net, ds, params, C=1.0):
    """One batched forward + loss + backward of the dataset through the C ABI against the C oracle: per-graph loss
    (== -cut of the partition the kernels chose), every gradient entry <= 1e-4 of the largest.  The workspace and
    the gradient buffer are poisoned first: nothing may be read before it is written in the same step.  Returns
    (engine, kernel tags of the step) so that callers can assert which kernel sequence ran."""
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], [it[0].edge_values(it[1]) for it in items], eng.device)
    eng.train_fwd_bwd(batch, C)        # sizes the workspace ...
    eng._ws.fill_(255)                 # ... which is then poisoned (all-ones bytes = NaN)
    eng.grad.fill_(float("nan"))
    with pkg.hip.Probe(64) as probe:
        P, S, loss = eng.train_fwd_bwd(batch, C)
    tags = [t for t, _ms in probe.records]
    ct = CO.CTrainer(params, Cc=C)
    csrs = csrs_of(ds)
    ref_loss = ct.step(csrs)
    loss_np, S_np = loss.cpu().numpy(), S.cpu().numpy()
    if not np.array_equal(loss_np, ref_loss):
        # A graph whose loss differs decoded some row differently.  That is legitimate only on a near-tie (the
        # summation order of the kernels is not the oracle's: top-2 margin inside fp32 noise); the gradient of
        # such a graph is then the oracle's backward for the partition the KERNELS chose, which is what is built
        # here: same forward, GP from the kernels' S, same backward.
        W = [params[k] for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias")]
        acc = [np.zeros_like(w) for w in W]
        off = 0
        for i, (rp, cl, vl) in enumerate(csrs):
            n = len(rp) - 1
            f = CO.forward(rp, cl, vl, *W)
            ref_s = f["P"].argmax(1); ref_s[:3] = [0, 1, 2]
            s_i = S_np[off:off + n]
            diff = np.nonzero(s_i != ref_s)[0]
            if diff.size == 0:
                assert loss_np[i] == ref_loss[i], i
            else:
                srt = np.sort(f["P"][diff].astype(np.float64), axis=1)
                assert (srt[:, 2] - srt[:, 1]).max() < 1e-6, (i, diff, srt)
            wv = np.ones(len(cl), np.float32) if vl is None else vl
            rows = np.repeat(np.arange(n), np.diff(rp))
            GP = np.zeros((n, 3), np.float32)
            np.add.at(GP, (rows, s_i[cl]), C * wv)                  # GP = C * A_val @ onehot(S)
            cut = 0.5 * float(wv[s_i[rows] != s_i[cl]].sum())
            assert loss_np[i] == np.float32(-C * cut), i            # loss == -C cut of the partition the kernels chose
            for a_, d_ in zip(acc, CO.backward(rp, cl, vl, W[0].shape[0], W[2], f["H"], f["P"], GP)):
                a_ += d_
            off += n
        ref = dict(zip(("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"), [a_.ravel() for a_ in acc]))
    else:
        o = np.cumsum([0, ct.N * ct.F, ct.F, ct.F * ct.K, ct.K])
        ref = {k: ct.grad[o[i]:o[i + 1]] for i, k in enumerate(("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"))}
    assert float(eng.grad[eng.count]) == float(loss_np.sum())   # GMC_MODEL_GRAD_TAIL: the loss rides behind the gradient
    for k, g in eng.views(eng.grad).items():
        g, r = g.cpu().numpy().ravel(), ref[k]
        assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), k
    # probabilities of every graph against the oracle forward
    off = 0
    Pn = P.cpu().numpy()
    Wl = [params[k] for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias")]
    for (rp, cl, vl) in csrs:
        n = len(rp) - 1
        assert np.abs(Pn[off:off + n] - CO.forward(rp, cl, vl, *Wl)["P"]).max() < 1e-4
        off += n
    return eng, tags
