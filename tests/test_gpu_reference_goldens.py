"""GPU tests that tie the HIP kernels to the fixtures the reference's OWN code produced
(tests/golden/training.json, see tests/golden/make_golden_training.py) and that run BASELINE.json's
configurations literally against the CPU oracles.  Everything goes through the C ABI."""
import json
import os

import networkx as nx
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R
from tests import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pkg(built):
    built.hip.require_gpu()
    return built


@pytest.fixture(scope="module")
def G():
    return json.load(open(os.path.join(HERE, "golden", "training.json")))


def graph_of(n, edges):
    g = nx.Graph()
    g.add_nodes_from(range(n))
    for u, v, w in edges:
        g.add_edge(int(u), int(v), weight=int(w), capacity=1)
    return g


def run_head(pkg, batch, Z0, b2, Cc, backward=True):
    lib, p = pkg.hip.load(), pkg.hip.ptr
    R_ = batch.R
    P = torch.full((R_, 3), float("nan"), device="cuda")
    S = torch.full((R_,), -1, dtype=torch.int32, device="cuda")
    loss = torch.full((batch.B,), float("nan"), device="cuda")
    db2 = torch.full((batch.B, 3), float("nan"), device="cuda") if backward else None
    GY2 = torch.full((R_, 4), float("nan"), device="cuda") if backward else None
    z = torch.from_numpy(np.ascontiguousarray(Z0, np.float32)).cuda()
    b = torch.from_numpy(np.ascontiguousarray(b2, np.float32)).cuda()
    pkg.hip.check(lib.gmc_head_f32(batch.ref(), p(z), 1, p(b), float(Cc), p(P), p(S), p(loss), p(GY2), p(db2),
                                   pkg.hip.stream()), "gmc_head_f32")   # (..., loss, GY2, db2part, stream)
    torch.cuda.synchronize()
    out = [P.cpu().numpy(), S.cpu().numpy(), loss.cpu().numpy()]
    return out + ([db2.cpu().numpy(), GY2.cpu().numpy()] if backward else [])


def adjacency64(g, weighted):
    n = g.number_of_nodes()
    a = np.zeros((n, n))
    for u, v, w in g.edges(data="weight", default=1):
        a[u, v] = a[v, u] = (w if weighted else 1.0)
    return a


@pytest.mark.parametrize("idx", range(5))
def test_head_kernel_against_reference_loss_chain(pkg, G, idx):
    """gmc_head_f32 (softmax, terminal override, argmax, cut loss, start of backward) against what the
    reference's override_fixed_nodes -> apply_max_to_one_hot -> compute_loss -> .backward() produced
    (TrainingNeural.py:87-106,154-176,291-309) for the same probabilities.

    The kernel takes pre-aggregation logits, so the fixture's P is injected by solving
    dinv o (A @ Z0) = log P in float64 (the rows the reference decided by its first-max rule on an exact
    tie get their winner nudged by 1e-3 so that fp32 noise cannot flip them; the exact-tie behaviour of the
    kernel itself is the next test).  Checked: S == the reference's argmax for every row, loss == the
    reference's loss, and db2 / GY2 == the softmax backward + aggregation of the REFERENCE's dL/dP."""
    c = G["loss"][idx]
    n, Cc = c["n"], c["C"]
    g = graph_of(n, c["edges"])
    h = pkg.from_networkx(g)
    batch = pkg.GraphBatch([h], None)
    P_ref = np.asarray(c["P"], np.float64)
    arg = np.asarray(c["argmax"])
    logp = np.log(np.maximum(P_ref, 1e-30))
    logp = np.maximum(logp, -30.0)
    logp[np.arange(n), arg] += 1e-3                      # break exact ties the way the reference did
    logp[:3] = np.log(np.maximum(P_ref[:3], 1e-30)).clip(-30)   # terminals: the override decides, not the row
    A1 = adjacency64(g, False)
    dinv = 1.0 / np.sqrt(A1.sum(1))
    Z0 = np.linalg.solve(dinv[:, None] * A1, logp)
    Z0f = Z0.astype(np.float32)
    z64 = dinv[:, None] * (A1 @ Z0f.astype(np.float64))
    e = np.exp(z64 - z64.max(1, keepdims=True))
    P64 = e / e.sum(1, keepdims=True)
    srt = np.sort(P64[3:], axis=1)
    assert (srt[:, 2] - srt[:, 1]).min() > 1e-5, "injected logits leave a row undecided"

    P, S, loss, db2, GY2 = run_head(pkg, batch, Z0f, np.zeros(3, np.float32), Cc)
    assert np.abs(P - P64).max() < 1e-4           # (the injected logits are large: solve conditioning)
    assert S.tolist() == c["argmax"]
    assert abs(float(loss[0]) - c["loss"]) <= 1e-6 * abs(c["loss"])
    assert float(loss[0]) == -Cc * R.cut_value(c["argmax"], g)
    # start of backward from the reference's own dL/dP
    GP = np.asarray(c["dL_dP"], np.float64)
    Pk = P.astype(np.float64)
    GZ = Pk * (GP - (GP * Pk).sum(1, keepdims=True))
    ref_db2 = GZ.sum(0)
    ref_gy2 = A1 @ (dinv[:, None] * GZ)
    scale = max(1.0, np.abs(ref_gy2).max())
    assert np.abs(db2[0] - ref_db2).max() <= 2e-5 * max(1.0, np.abs(ref_db2).max())
    assert np.abs(GY2[:, :3] - ref_gy2).max() <= 2e-5 * scale
    assert np.allclose(GY2[:, 3], dinv, rtol=1e-6)


def test_head_kernel_exact_ties_take_the_first_maximum(pkg):
    """TrainingNeural.py:98 (torch.argmax: first maximum).  Identical Z0 columns give bit-identical
    logits, hence bit-identical probabilities: classes {0,1} tie on some rows, {1,2} on others, all three
    on the rest - the kernel must pick the lowest class index, and the cut / loss must follow it."""
    g = R.regular_graph(60, 7, 77)
    h = pkg.from_networkx(g)
    batch = pkg.GraphBatch([h], None)
    n = 60
    rng = np.random.RandomState(5)
    base = rng.standard_normal(n).astype(np.float32)
    # a per-node column pattern would be mixed by the aggregation, so the ties are built on the aggregated
    # logits: all three columns equal everywhere, then b2 separates the classes the same way for every row
    Z0 = np.repeat(base[:, None], 3, axis=1).astype(np.float32)
    for b2, want in (((0.0, 0.0, 0.0), 0), ((0.0, 0.5, 0.5), 1), ((0.25, 0.25, 0.0), 0), ((-1.0, 0.0, 0.0), 1),
                     ((-1.0, -1.0, 0.0), 2)):
        P, S, loss = run_head(pkg, batch, Z0, np.asarray(b2, np.float32), 1.0, backward=False)
        top = P.max(1)
        ties = (P == top[:, None]).sum(1)
        assert (ties >= 1).all() and (ties == sum(1 for v in b2 if v == max(b2))).all()   # bit-identical maxima
        expect = np.full(n, want)
        expect[:3] = [0, 1, 2]
        assert S.tolist() == expect.tolist(), b2
        assert float(loss[0]) == -R.cut_value(expect.tolist(), g)


def test_reference_evaluate_model_fixture_on_the_gpu_head(pkg, G):
    """evaluate_model (TrainingNeural.py:537-570) of the reference on stored probabilities: total and
    average loss reproduced by the head kernel's per-graph losses on the same (injected) probabilities."""
    c = G["evaluate_model"][0]
    total = 0.0
    for gr in c["graphs"]:
        n = gr["n"]
        g = graph_of(n, gr["edges"])
        batch = pkg.GraphBatch([pkg.from_networkx(g)], None)
        P_ref = np.asarray(gr["P"], np.float64)
        # the reference's decode on these rows (first max on ties), reproduced by nudging as above
        arg = P_ref.argmax(1)
        logp = np.log(np.maximum(P_ref, 1e-30)).clip(-30)
        logp[np.arange(n), arg] += 1e-3
        A1 = adjacency64(g, False)
        dinv = 1.0 / np.sqrt(A1.sum(1))
        Z0 = np.linalg.solve(dinv[:, None] * A1, logp).astype(np.float32)
        _, S, loss = run_head(pkg, batch, Z0, np.zeros(3, np.float32), 1.0, backward=False)
        total += float(loss[0])
    assert total == c["result"]["total_loss"] and total / len(c["graphs"]) == c["result"]["average_loss"]
    assert c["result"]["num_samples"] == len(c["graphs"])


# --------------------------------------------------------------------------- BASELINE configs, literally
def model(pkg, hidden, seed=0):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=hidden)
    torch.manual_seed(seed)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    return T, cfg, net, embed, opt, util.np_params(net.state_dict())


def flat_ref_grads(ct):
    o = np.cumsum([0, ct.N * ct.F, ct.F, ct.F * ct.K, ct.K])
    return {"conv1.weight": ct.grad[o[0]:o[1]], "conv1.bias": ct.grad[o[1]:o[2]],
            "conv2.weight": ct.grad[o[2]:o[3]], "conv2.bias": ct.grad[o[3]:o[4]]}


def check_step_against_c_oracle(pkg, specs, hidden=500):
    """One batched training step (forward, loss, backward) of the default fused sequence vs the scalar C
    oracle: probabilities 1e-4, argmax exact where the float64-free margin is clear, per-graph losses, and the
    batch gradient 1e-4 of its largest entry (where a graph decodes differently on a sub-1e-6 margin the
    gradient differs by design: then the differing rows must all be near-ties)."""
    T, cfg, net, embed, opt, params = model(pkg, hidden)
    ds = util.product_dataset(specs)
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    eng.train_fwd_bwd(batch, 1.0)
    eng._ws.fill_(255)
    eng.grad.fill_(float("nan"))
    P, S, loss = eng.train_fwd_bwd(batch, 1.0)
    P, S, loss = P.cpu().numpy(), S.cpu().numpy(), loss.cpu().numpy()
    csrs = util.csrs_of(ds)
    ct = CO.CTrainer(params)
    ref_loss = ct.step(csrs)
    off, undecided = 0, 0
    for i, (rp, cl, vl) in enumerate(csrs):
        n = len(rp) - 1
        f = CO.forward(rp, cl, vl, params["conv1.weight"], params["conv1.bias"], params["conv2.weight"], params["conv2.bias"])
        assert np.abs(P[off:off + n] - f["P"]).max() < 1e-4
        ref_s = f["P"].argmax(1)
        ref_s[:3] = [0, 1, 2]
        diff = np.nonzero(S[off:off + n] != ref_s)[0]
        if diff.size:   # only rows whose top-2 margin is inside fp32 noise may decode differently
            srt = np.sort(f["P"][diff].astype(np.float64), axis=1)
            assert (srt[:, 2] - srt[:, 1]).max() < 1e-6, (i, diff, srt)
            undecided += diff.size
        else:
            assert loss[i] == ref_loss[i], i
        off += n
    assert float(eng.grad[eng.count]) == float(loss.sum())
    if undecided == 0:
        # relu kinks: among the batch's millions of layer-1 pre-activations a few land within fp32 rounding
        # of 0, kernels and oracle may take different sides, and then that hidden column's dW1 column and db1
        # entry differ by one row's contribution.  Such columns are identified in float64 and are the ONLY
        # place where the 1e-4 bar may be missed.
        kink = np.zeros(hidden, bool)
        W1, b1 = params["conv1.weight"].astype(np.float64), params["conv1.bias"].astype(np.float64)
        for (_g, _a, nx_g, _t) in items:
            n = nx_g.number_of_nodes()
            a = np.zeros((n, n))
            for u, v in nx_g.edges():
                a[u, v] = a[v, u] = 1.0
            dis = 1.0 / np.sqrt(a.sum(1))
            pre = dis[:, None] * (a @ (dis[:, None] * (a @ W1[:n]))) + b1
            kink |= (np.abs(pre) < 3e-8).any(0)     # (sums of ~50 terms of ~3e-3: fp32 accumulation noise ~1e-8)
        ref = flat_ref_grads(ct)
        bad_cols = set()
        for k, gr in eng.views(eng.grad).items():
            gr, r = gr.cpu().numpy(), ref[k].reshape(tuple(gr.shape))
            tol = 1e-4 * max(1.0, np.abs(r).max())
            d = np.abs(gr - r)
            if k == "conv1.weight":
                bad_cols |= set(np.nonzero(d.max(0) > tol)[0])
            elif k == "conv1.bias":
                bad_cols |= set(np.nonzero(d > tol)[0])
            else:
                assert d.max() <= tol, k
        assert bad_cols <= set(np.nonzero(kink)[0]), (sorted(bad_cols), np.nonzero(kink)[0])
        assert len(bad_cols) <= max(2, len(items) // 10)
    return undecided


def test_config3_full_size_step_against_the_c_oracle(pkg):
    """BASELINE configs[3] at full size on one GPU: 160 x n=1000 d=7 graphs, hidden 500 - the bench
    workload itself, every graph against the C oracle (P, argmax, loss) and the 502,003-entry batch
    gradient."""
    specs = [(1000, 7, 3000 + i) for i in range(160)]
    und = check_step_against_c_oracle(pkg, specs)
    assert und <= 2


def test_config1_twenty_n500_graphs_literally(pkg):
    """BASELINE configs[1]: batch of 20 x n=500 regular graphs with d = 6..8 (C2,
    seeds 1000+i, d_i = 6 + i mod 3)."""
    specs = [(500, 6 + (i % 3), 1000 + i) for i in range(20)]
    und = check_step_against_c_oracle(pkg, specs)
    assert und <= 1


def test_config4_mixed_sizes_inference_with_200_iteration_post_processing(pkg):
    """BASELINE configs[4]: n in {50,100,200,300,500} inference + the reference's 200-iteration
    post-processing (Testing/TestingNeuralNetwork.py:66-98) - the GPU sampler with host uniforms in the
    reference's draw order against the oracle restatement (itself pinned against the reference's own
    post_processing_optimization by tests/golden/decode.json), bit for bit: same best assignment, same cut."""
    from gcn_max_cut_amd.Testing import TestingNeuralNetwork as TN
    T, cfg, net, *_ = model(pkg, 500)
    specs = [(n, 6 + (i % 3), n * 1000 + i) for n in (50, 100, 200, 300, 500) for i in range(2)]
    ds = util.product_dataset(specs)
    net.eval()
    for key, (g, a_pad, nx_g, _t) in ds.items():
        with torch.no_grad():
            probs = net(g, a_pad)
        p_host = probs.cpu().numpy()
        np.random.seed(key)
        got, got_cut = TN.post_processing_optimization(probs, nx_g, 200)
        np.random.seed(key)
        ref, ref_cut = R.post_process(p_host, nx_g, 200, np.random.rand)
        assert list(got) == list(ref) and got_cut == ref_cut, key
        simple = TN.simple_partition_assignment(probs)
        assert TN.calculate_cut_value(simple, nx_g) == R.cut_value(list(R.partition_of(torch.from_numpy(p_host))), nx_g)


def test_evaluate_optimal_partitioning_on_the_gpu_model(pkg):
    """GetOptimalNetValue / evaluate_optimal_partitioning (TrainingNeural.py:253-289) on the real HIP
    model, against the oracle restatement of the same lines fed with the oracle's probabilities: six
    forwards (the permutations never reach the model), terminal override, 0.5 threshold, legacy cut."""
    T, cfg, net, embed, opt, params = model(pkg, 64)
    ds = util.product_dataset([(40, 5, 41), (64, 6, 43)])
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    for (g, a_pad, nx_g, _t) in ds.values():
        adj = a_pad[:, :g.number_of_nodes()].cpu()
        got = T.GetOptimalNetValue(net, g, a_pad, adj.cuda(), {"s": 0, "t": 1, "u": 2})
        P = R.forward(tp, R.graph_from_networkx(nx_g), a_pad.cpu())
        binary = (R.override_terminals(P) >= 0.5).float()
        # restatement of :229-251 (sum over classes of the weight between members and non-members, / 2 / 2)
        want = 0.0
        for k in range(3):
            col = binary[:, k]
            want += float((adj * (col[None, :] != col[:, None]).float()).sum() / 2)
        want /= 2
        srt = np.sort(P.numpy(), axis=1)
        assert np.abs(srt - 0.5).min() > 1e-4          # no probability on the threshold
        assert float(got) == want
