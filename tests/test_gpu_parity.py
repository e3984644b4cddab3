"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances: node probabilities 1e-4 (north star), argmax partitions exact,
loss == -cut exactly, gradients 1e-4 relative to the largest entry."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R
from tests import util

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4


@pytest.fixture(scope="module")
def pkg(built):
    built.hip.require_gpu()
    return built


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_spmm(pkg, rp, cl, vl, scale, X, bias, relu, group_rows=0, W2=None):
    lib = pkg.hip.load()
    n, F = rp.size - 1, X.shape[1]
    d = {k: (None if v is None else dev(v)) for k, v in
         dict(rp=rp, cl=cl, vl=vl, scale=scale, X=X, bias=bias, W2=W2).items()}
    Y = torch.full((n, F), float("nan"), device="cuda")
    Z0 = torch.full((n, 3), float("nan"), device="cuda") if W2 is not None else None
    p = pkg.hip.ptr
    rc = lib.gmc_spmm_f32(p(d["rp"]), p(d["cl"]), p(d["vl"]), p(d["scale"]), p(d["X"]), F, p(d["bias"]),
                          int(relu), p(Y), F, n, F, group_rows, p(d["W2"]), p(Z0), pkg.hip.stream())
    pkg.hip.check(rc, "gmc_spmm_f32")
    torch.cuda.synchronize()
    return Y.cpu().numpy(), None if Z0 is None else Z0.cpu().numpy()


def block_diag_csr(specs):
    """Block-diagonal CSR (global columns) of seeded regular graphs."""
    rps, cls, off, eoff = [np.zeros(1, np.int32)], [], 0, 0
    for n, d, s in specs:
        rp, cl, _ = CO.csr_of(R.regular_graph(n, d, s))
        rps.append(rp[1:] + eoff)
        cls.append(cl + off)
        off += n
        eoff += cl.size
    return np.concatenate(rps).astype(np.int32), np.concatenate(cls).astype(np.int32)


@pytest.mark.parametrize("F", [500, 256, 64, 16, 1000, 6])
def test_spmm_pure_sum_is_bit_exact(pkg, F):
    rp, cl = block_diag_csr([(100, 7, 1), (60, 6, 2), (80, 8, 3)])
    rng = np.random.RandomState(F)
    X = rng.standard_normal((rp.size - 1, F)).astype(np.float32)
    ref = CO.spmm(rp, cl, None, None, X, None, False)
    got, _ = run_spmm(pkg, rp, cl, None, None, X, None, False)
    assert np.array_equal(got, ref)  # same CSR summation order, no scaling


@pytest.mark.parametrize("F,group", [(500, 0), (500, 100), (128, 60), (12, 0)])
def test_spmm_epilogues(pkg, F, group):
    rp, cl = block_diag_csr([(100, 7, 4)] * 9 + [(60, 6, 5)])
    n = rp.size - 1
    rng = np.random.RandomState(7)
    X = rng.standard_normal((n, F)).astype(np.float32)
    vl = rng.randint(1, 4, cl.size).astype(np.float32)
    scale = rng.rand(n).astype(np.float32) + 0.5
    bias = rng.standard_normal(F).astype(np.float32)
    ref = CO.spmm(rp, cl, vl, scale, X, bias, True)
    got, _ = run_spmm(pkg, rp, cl, vl, scale, X, bias, True, group)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)


def test_spmm_fused_w2_epilogue_and_mfma_kernel(pkg):
    rp, cl = block_diag_csr([(200, 7, 8), (120, 6, 9)])
    n, F = rp.size - 1, 500
    rng = np.random.RandomState(3)
    X = rng.standard_normal((n, F)).astype(np.float32)
    scale = (1.0 / np.sqrt(np.diff(rp))).astype(np.float32)
    bias = rng.standard_normal(F).astype(np.float32) * 0.1
    W2 = (rng.rand(F, 3).astype(np.float32) - 0.5) * 0.2
    H, Z0 = run_spmm(pkg, rp, cl, None, scale, X, bias, True, 0, W2)
    ref = (H.astype(np.float64) * scale[:, None]) @ W2.astype(np.float64)
    np.testing.assert_allclose(Z0, ref, rtol=1e-5, atol=1e-5)
    # the matrix-core form of the same contraction
    lib = pkg.hip.load()
    Hd, sd, Wd, Zm = dev(H), dev(scale), dev(W2), torch.empty((n, 3), device="cuda")
    p = pkg.hip.ptr
    rc = lib.gmc_dense_hw2_f32(p(Hd), F, p(sd), p(Wd), p(Zm), n, F, pkg.hip.stream())
    pkg.hip.check(rc, "gmc_dense_hw2_f32")
    np.testing.assert_allclose(Zm.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_mfma_dense_integer_exact(pkg):
    """A=I-style check with asymmetric integer data: exact in fp32, catches layout swaps."""
    n, F = 37, 48
    H = (np.arange(n * F).reshape(n, F) % 7 - 3).astype(np.float32)
    W2 = (np.arange(F * 3).reshape(F, 3) % 5 - 2).astype(np.float32)
    d = np.ones(n, np.float32) * 2
    lib = pkg.hip.load()
    Z, Hd, dd, Wd = torch.empty((n, 3), device="cuda"), dev(H), dev(d), dev(W2)
    p = pkg.hip.ptr
    pkg.hip.check(lib.gmc_dense_hw2_f32(p(Hd), F, p(dd), p(Wd), p(Z), n, F, pkg.hip.stream()), "mfma")
    assert np.array_equal(Z.cpu().numpy(), 2 * (H @ W2))


def model_and_params(pkg, hidden, seed=0):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=hidden)
    torch.manual_seed(seed)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    return T, cfg, net, embed, opt, util.np_params(net.state_dict())


SPECS_SMALL = [(100, 7, 1000), (50, 6, 1001), (64, 8, 1002), (30, 5, 1003)]


@pytest.mark.parametrize("hidden,specs", [(16, SPECS_SMALL), (500, [(500, 7, 1000)]), (500, [(1000, 7, 2000)]),
                                          (128, [(200, 6, 7), (300, 8, 8)]),
                                          # degrees 9..16: the 16-slot neighbour table; 20: no table (row kernels)
                                          (64, [(120, 12, 11), (90, 10, 12), (60, 9, 13)]), (32, [(80, 20, 14)])])
def test_forward_probabilities_and_partitions(pkg, hidden, specs):
    T, cfg, net, *_rest, params = model_and_params(pkg, hidden)
    ds = util.product_dataset(specs)
    net.eval()
    for (g, a_pad, nx_g, _t) in ds.values():
        with torch.no_grad():
            P = net(g, a_pad).cpu().numpy()
        rp, cl, vl = CO.csr_of(nx_g)
        ref = CO.forward(rp, cl, vl, params["conv1.weight"], params["conv1.bias"], params["conv2.weight"],
                         params["conv2.bias"])["P"]
        assert np.abs(P - ref).max() < PROB_TOL
        # reference-structured dense oracle (dense [n,1000] GEMM, DGL order)
        tp = {k: torch.from_numpy(v) for k, v in params.items()}
        og = R.graph_from_networkx(nx_g)
        Pd = R.forward(tp, og, a_pad.cpu()).numpy()
        assert np.abs(P - Pd).max() < PROB_TOL
        # argmax partitions: exact wherever the float64 margin is not inside fp32 noise
        P64 = R.forward_dense_f64(tp, nx_g)
        srt = np.sort(P64, axis=1)
        decided = (srt[:, 2] - srt[:, 1]) > 1e-6
        assert decided.mean() > 0.99
        assert np.array_equal(P.argmax(1)[decided], P64.argmax(1)[decided])
        # against the C oracle: every row decoded differently must be a float64 near-tie (margin < 1e-6)
        differ = np.nonzero(P.argmax(1) != ref.argmax(1))[0]
        assert ((srt[differ, 2] - srt[differ, 1]) < 1e-6).all(), (differ, srt[differ])


def test_batched_forward_is_bitwise_the_per_graph_forward(pkg):
    """A graph's probabilities do not depend on which graphs share its batch or where it sits in it
    (bitwise), and are the single-graph forward exactly when the launch shape class is the same
    (same LDS slice width and slices per group); across classes the fused H@W2 partials are folded in
    a different association: equal to rounding."""
    T, cfg, net, *_ = model_and_params(pkg, 500)
    eng = net.engine()
    big = [(1000, 7, 3000 + i) for i in range(9)]
    items = list(util.product_dataset(big).values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = [t.clone() for t in eng.forward(batch, 1.0, want_loss=True)]
    # same graphs, reversed order, two of them swapped for others: per-graph results unchanged
    other = list(util.product_dataset([(1000, 7, 77), (1000, 7, 78)]).values())
    perm = items[::-1][:7] + other
    batch2 = pkg.GraphBatch([it[0] for it in perm], None, eng.device)
    P2, S2, loss2 = eng.forward(batch2, 1.0, want_loss=True)
    for j in range(7):
        g = len(items) - 1 - j
        assert torch.equal(batch.split(P)[g], batch2.split(P2)[j])
        assert torch.equal(batch.split(S)[g], batch2.split(S2)[j])
        assert float(loss[g]) == float(loss2[j])
    # single graphs: exact with the batch's grouping forced, to rounding with their own
    for g, it in enumerate(items[:3]):
        single = pkg.GraphBatch([it[0]], None, eng.device)
        Pg, Sg, lg = eng.forward(single, 1.0, want_loss=True)
        assert float((batch.split(P)[g] - Pg).abs().max()) < 1e-6
    # same class (4 slices per group from 16 graphs of this size on): batch of 40 == batch of 20
    many = list(util.product_dataset([(1000, 7, 5000 + i) for i in range(40)]).values())
    b40 = pkg.GraphBatch([it[0] for it in many], None, eng.device)
    b20 = pkg.GraphBatch([it[0] for it in many[10:30]], None, eng.device)
    P40 = eng.forward(b40, 1.0)[0].clone()
    P20 = eng.forward(b20, 1.0)[0]
    for j in range(20):
        assert torch.equal(b40.split(P40)[10 + j], b20.split(P20)[j])
    mixed = [(1000, 7, 3000), (500, 6, 1), (300, 8, 2), (64, 5, 3)]
    items = list(util.product_dataset(mixed).values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = eng.forward(batch, 1.0, want_loss=True)
    for g, it in enumerate(items):
        single = pkg.GraphBatch([it[0]], None, eng.device)
        Pg, Sg, lg = eng.forward(single, 1.0, want_loss=True)
        assert float((batch.split(P)[g] - Pg).abs().max()) < 1e-6


def test_graphs_larger_than_the_lds_tiles_use_the_row_kernels(pkg):
    """n_nodes = 1600: graphs beyond the LDS tile capacity (n > 1020) take the row-per-wave kernels;
    same parity bar (the C oracle), mixed in one batch with a graph that would fit."""
    from gcn_max_cut_amd.Training import TrainingNeural as T
    cfg = T.TrainingConfig(n_nodes=1600, hidden_dim=48)
    torch.manual_seed(1)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    params = util.np_params(net.state_dict())
    specs = [(1500, 7, 61), (1100, 6, 62), (200, 8, 63)]
    ds = util.product_dataset(specs, max_nodes=1600)
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = eng.train_fwd_bwd(batch, 1.0)
    ct = CO.CTrainer(params)
    ref_loss = ct.step(util.csrs_of(ds))
    assert np.array_equal(loss.cpu().numpy(), ref_loss)
    ref = flat_ref_grads(ct)
    for k, g in eng.views(eng.grad).items():
        g, r = g.cpu().numpy().ravel(), ref[k]
        assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), k
    off = 0
    for (rp, cl, vl) in util.csrs_of(ds):
        f = CO.forward(rp, cl, vl, params["conv1.weight"], params["conv1.bias"], params["conv2.weight"], params["conv2.bias"])
        n = len(rp) - 1
        assert np.abs(P[off:off + n].cpu().numpy() - f["P"]).max() < PROB_TOL
        off += n


def test_persistent_forward_ranges_cross_graphs(pkg, monkeypatch):
    """The fused forward hands every workgroup a contiguous range of (graph, slice group) items.
    With 4 'CUs' a range spans several graphs of different sizes: the result must be bitwise the
    one-item-per-workgroup result (and within tolerance of the oracle, checked elsewhere)."""
    T, cfg, net, *_ = model_and_params(pkg, 500)
    eng = net.engine()
    specs = [(300, 7, 41), (120, 6, 42), (1000, 7, 43), (64, 5, 44), (500, 8, 45), (250, 7, 46), (90, 6, 47)]
    items = list(util.product_dataset(specs).values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    lib = pkg.hip.load()
    try:
        lib.gmc_debug_set_device_cus(100000)   # one item per workgroup
        P1, S1, l1 = [t.clone() for t in eng.forward(batch, 1.0, want_loss=True)]
        for cus in (4, 3, 1):
            lib.gmc_debug_set_device_cus(cus)
            P2, S2, l2 = eng.forward(batch, 1.0, want_loss=True)
            assert torch.equal(P1, P2) and torch.equal(S1, S2) and torch.equal(l1, l2), cus
        lib.gmc_debug_set_device_cus(3)
        eng.train_fwd_bwd(batch, 1.0)
        g3 = eng.grad.clone()
    finally:
        lib.gmc_debug_set_device_cus(0)
    eng.train_fwd_bwd(batch, 1.0)
    assert torch.equal(g3, eng.grad)


def test_loss_equals_minus_cut_of_argmax_partition(pkg):
    T, cfg, net, *_ = model_and_params(pkg, 64)
    ds = util.product_dataset(SPECS_SMALL + [(500, 7, 5)])
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = eng.forward(batch, 1.0, want_loss=True)
    for g, it in enumerate(items):
        part = batch.split(S)[g].cpu().numpy()
        assert list(part[:3]) == [0, 1, 2]
        assert np.array_equal(part, R.partition_of(batch.split(P)[g].cpu()))
        assert float(loss[g]) == -float(R.cut_value(part.tolist(), it[2]))
    res = T.evaluate_model(net, ds, cfg)
    assert res["num_samples"] == len(items)
    assert res["total_loss"] == float(loss.sum())


def test_full_size_batch_properties(pkg):
    """BASELINE configs[3] at full size (160 graphs, n=1000, d=7, hidden 500) through properties that
    need no oracle run: rows of P sum to 1, terminals keep their classes, loss == -cut(argmax) exactly
    (integers), node relabelling permutes P (to rounding), rows of dW1 beyond n stay zero, and the
    step is bitwise reproducible."""
    T, cfg, net, *_ = model_and_params(pkg, 500)
    eng = net.engine()
    B = 160
    items = list(util.product_dataset([(1000, 7, 7000 + i) for i in range(B - 1)] + [(600, 7, 7999)]).values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = [t.clone() for t in eng.train_fwd_bwd(batch, 1.0)]
    g1 = eng.grad.clone()
    assert float((P.sum(1) - 1).abs().max()) < 1e-6
    Sh, lh = S.cpu().numpy(), loss.cpu().numpy()
    off = 0
    for g, it in enumerate(items):
        n = it[2].number_of_nodes()
        part = Sh[off:off + n]
        assert list(part[:3]) == [0, 1, 2]
        if g % 16 == 0 or g == B - 1:
            assert float(lh[g]) == -float(R.cut_value(part.tolist(), it[2]))
        off += n
    assert float(lh.sum()) == float(sum(lh.tolist())) and np.all(lh == np.round(lh))
    eng._ws.fill_(255)
    P2, S2, loss2 = eng.train_fwd_bwd(batch, 1.0)
    assert torch.equal(P, P2) and torch.equal(S, S2) and torch.equal(loss, loss2) and torch.equal(g1, eng.grad)
    # relabel the non-terminal nodes of one graph: probabilities follow the permutation
    import networkx as nx
    g0 = items[5][2]
    n = g0.number_of_nodes()
    perm = np.arange(n); rng = np.random.RandomState(3); perm[3:] = 3 + rng.permutation(n - 3)
    h = nx.relabel_nodes(g0, {int(u): int(perm[u]) for u in g0.nodes()})
    hs = nx.Graph(); hs.add_nodes_from(range(n)); hs.add_edges_from(h.edges(data=True))
    one = pkg.GraphBatch([pkg.from_networkx(g0)], None, eng.device)
    two = pkg.GraphBatch([pkg.from_networkx(hs)], None, eng.device)
    Pa = eng.forward(one, 1.0)[0].clone()
    Pb = eng.forward(two, 1.0)[0]
    # features = adjacency rows, so W1's rows are tied to node ids: equivariance holds for the
    # aggregation structure only when W1 is permuted alongside; check that instead
    with torch.no_grad():
        W1 = net.conv1.weight.clone()
        inv = torch.from_numpy(perm).to(W1.device)
        net.conv1.weight[:n] = W1[:n][torch.argsort(inv)]
    Pc = eng.forward(two, 1.0)[0].clone()
    with torch.no_grad():
        net.conv1.weight.copy_(W1)
    assert float((Pc[inv] - Pa).abs().max()) < 1e-5
    assert float((Pb[inv] - Pa).abs().max()) > 1e-5   # (and without it the outputs genuinely differ)


def flat_ref_grads(ct):
    o = np.cumsum([0, ct.N * ct.F, ct.F, ct.F * ct.K, ct.K])
    return {k: ct.grad[o[i]:o[i + 1]] for i, k in enumerate(("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"))}


@pytest.mark.parametrize("hidden,specs", [(16, SPECS_SMALL), (500, [(500, 7, 1000), (500, 6, 1001)]),
                                          (500, [(1000, 7, 2000 + i) for i in range(9)]),
                                          # the reference's default dataset name is ..._d8_12_...: degrees up to 12
                                          (64, [(120, 12, 11), (90, 10, 12), (60, 9, 13), (200, 8, 15)]),
                                          (500, [(300, 12, 21), (250, 11, 22), (200, 9, 23)]),
                                          (32, [(80, 20, 14), (50, 7, 16)]),
                                          # 8 rows per thread (see test_one_kernel_per_operation_sequence_matches_oracle)
                                          (128, [(530, 7, 31), (520, 6, 32)]), (256, [(270, 7, 33), (262, 8, 34)])])
def test_step_gradients_match_oracle(pkg, hidden, specs):
    T, cfg, net, embed, opt, params = model_and_params(pkg, hidden)
    ds = util.product_dataset(specs)
    eng, _tags = util.check_step_against_oracle(pkg, net, ds, params)
    # rows of dW1 that no graph reaches are exactly zero
    nmax = max(s[0] for s in specs)
    assert float(eng.views(eng.grad)["conv1.weight"][nmax:].abs().max() if nmax < 1000 else 0.0) == 0.0


def test_step_gradients_match_torch_autograd_oracle(pkg):
    T, cfg, net, embed, opt, params = model_and_params(pkg, 16)
    ds = util.product_dataset(SPECS_SMALL)
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    eng.train_fwd_bwd(batch, 1.0)
    ods = util.oracle_dataset(SPECS_SMALL)
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    _, grads, _ = R.loss_and_grads(tp, list(ods.values()))
    for k, g in eng.views(eng.grad).items():
        r = grads[k].numpy()
        assert np.abs(g.cpu().numpy() - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), k


@pytest.mark.parametrize("fuse,specs", [(1, [(80, 7, 21), (40, 6, 22)]), (0, [(80, 7, 21), (40, 6, 22)]),
                                        (1, [(600, 7, 23), (90, 11, 24)]), (0, [(600, 7, 23), (90, 11, 24)])])
def test_weighted_edges_and_loss_scale(pkg, fuse, specs):
    """Non-unit edge weights (the `vals` / `ell_vals` variants of every kernel), C != 1, in the fused
    default and in the one-kernel-per-operation sequence."""
    T, cfg, net, embed, opt, params = model_and_params(pkg, 32)
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    graphs, terms = util.weighted_copy(specs)
    ds = GE.process_graphs_from_folder(graphs, terms, 1000)
    eng = net.engine()
    items = list(ds.values())
    vals = [it[0].edge_values(it[1]) for it in items]
    assert all(v is not None for v in vals)
    batch = pkg.GraphBatch([it[0] for it in items], vals, eng.device)
    lib = pkg.hip.load()
    prev = lib.gmc_set_fuse(fuse)
    try:
        P, S, loss = eng.train_fwd_bwd(batch, 2.5)
    finally:
        lib.gmc_set_fuse(prev)
    ct = CO.CTrainer(params, Cc=2.5)
    ref_loss = ct.step(util.csrs_of(ds))
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-6)
    ref = flat_ref_grads(ct)
    for k, g in eng.views(eng.grad).items():
        r = ref[k]
        assert np.abs(g.cpu().numpy().ravel() - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), k


def test_fused_adam_matches_torch_adam(pkg):
    lib = pkg.hip.load()
    n = 502003
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** float(step - 2))
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        hp = pkg.hip.ptr
        pkg.hip.check(lib.gmc_adam_f32(hp(p), hp(gd), hp(m), hp(v), n, 1e-3, 0.9, 0.999, 1e-8, step,
                                       pkg.hip.stream()), "adam")
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=3e-7, atol=3e-7)
    st = opt.state[ref]
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-6, atol=5e-7)
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("specs", [SPECS_SMALL, [(60, 20, 51), (100, 7, 52), (40, 12, 53)]])
def test_sequential_training_follows_oracle(pkg, specs):
    """Three epochs of the reference schedule (one Adam step per graph), replayed as a hipGraph;
    the second dataset mixes a degree-20 graph (row kernels, stand-alone Adam inside the captured
    step) with 8- and 16-slot ones."""
    T, cfg, net, embed, opt, params = model_and_params(pkg, 16)
    ds = util.product_dataset(specs)
    ct = CO.CTrainer(params, lr=cfg.learning_rate)
    csrs = util.csrs_of(ds)
    for epoch in range(3):
        got = T.train_single_epoch(ds, net, opt, embed, cfg)
        ref = float(sum(float(ct.step([c])[0]) for c in csrs))
        assert abs(got - ref) <= 0.02 * abs(ref) + 1e-6, (epoch, got, ref)
    after = ct.unpack()
    for k, v in net.state_dict().items():
        assert np.abs(v.cpu().numpy() - after[k]).max() < 5e-3, k  # a few +-lr Adam steps


def test_data_parallel_step_sequence_equals_the_fused_step(pkg):
    """The step a data-parallel rank runs (train_fwd_bwd -> [all-reduce] -> stand-alone Adam, loss in
    the gradient's tail slot) against the single-GPU fused step (fold + Adam in one sweep, replayed as
    a hipGraph): same parameters after a few epochs, same epoch losses."""
    import copy

    class WithoutFusedStep:          # an engine that offers only what the N > 1 branch uses
        def __init__(self, eng):
            self._eng = eng

        def __getattr__(self, name):
            if name == "train_step":
                raise AttributeError(name)
            return getattr(self._eng, name)

    specs = [(1000, 7, 81), (640, 6, 82), (300, 8, 83), (90, 11, 84)]
    ds = util.product_dataset(specs)
    runs = []
    for variant in ("fused", "dp-sequence"):
        T, cfg, net, embed, opt, params = model_and_params(pkg, 64, seed=5)
        eng = net.engine()
        tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=2,
                            engine=eng if variant == "fused" else WithoutFusedStep(eng))
        losses = [tr.epoch(ds) for _ in range(4)]
        assert (tr._graph is not None) == (variant == "fused")
        runs.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, int(eng.step_count)))
    (la, pa, sa), (lb, pb, sb) = runs
    assert sa == sb == 8 and la == lb
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) < 1e-6, k


def test_autograd_path_matches_fused_step(pkg):
    """net(g, A) -> reference-style helper chain -> .backward() uses the HIP backward."""
    T, cfg, net, embed, opt, params = model_and_params(pkg, 32)
    ds = util.product_dataset([(100, 7, 31)])
    (g, a_pad, nx_g, _t), = ds.values()
    net.train()
    P = net(g, a_pad)
    assert P.requires_grad
    s = T.apply_max_to_one_hot(T.override_fixed_nodes(P))
    loss = T.compute_loss(s, a_pad, cfg.A, cfg.C, cfg.penalty)
    opt.zero_grad()
    loss.backward()
    auto = {k: dict(net.named_parameters())[k].grad.clone() for k in pkg.engine.PARAM_ORDER}
    eng = net.engine()
    batch = pkg.GraphBatch([g], None, eng.device)
    _, _, fl = eng.train_fwd_bwd(batch, cfg.C)
    assert abs(float(loss.detach()) - float(fl[0])) < 1e-3
    for k, v in eng.views(eng.grad).items():
        assert np.abs((v - auto[k]).cpu().numpy()).max() <= 1e-5 * max(1.0, float(auto[k].abs().max())), k


def test_error_behaviour(pkg):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    import networkx as nx
    with pytest.raises(ValueError):
        pkg.FusedEngine(1000, 500, 2)  # 2-way model: the override is 3-wide
    g = nx.path_graph(5)
    g.add_node(5)  # isolated node
    with pytest.raises(pkg.DGLError):
        pkg.GraphBatch([pkg.from_networkx(g)], None)
    lib = pkg.hip.load()
    assert lib.gmc_spmm_f32(None, None, None, None, None, 4, None, 0, None, 4, 1, 4, 0, None, None, None) == -1
    assert lib.gmc_adam_f32(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 1, None) == -1
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=16)
    net, _, _ = T.setup_model_and_optimizer(cfg)
    big = pkg.from_networkx(nx.random_regular_graph(4, 1200, seed=1))
    with pytest.raises(ValueError):
        net.engine().forward(pkg.GraphBatch([big], None))  # more nodes than rows of conv1.weight


@pytest.mark.parametrize("hidden,specs", [(16, SPECS_SMALL), (500, [(1000, 7, 5), (960, 7, 6), (500, 6, 8)]),
                                          (64, [(120, 12, 11), (90, 10, 12)]),
                                          # 8 rows per thread: 32-column tiles for 512 < n <= 538, 64-column for 256 < n <= 275
                                          (128, [(530, 7, 31), (520, 6, 32)]), (256, [(270, 7, 33), (262, 8, 34)])])
def test_one_kernel_per_operation_sequence_matches_oracle(pkg, hidden, specs):
    """gmc_set_fuse(0): the stand-alone LDS SpMM / hidden-backward / dW1 kernels (the sequence whose
    SpMM bench.py's `roofline` times) against the C oracle, same bar as the fused default.
    (The relu-kink case (900, 7, 6) has its own test below.)"""
    T, cfg, net, embed, opt, params = model_and_params(pkg, hidden)
    ds = util.product_dataset(specs)
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    lib = pkg.hip.load()
    prev = lib.gmc_set_fuse(0)
    try:
        eng.train_fwd_bwd(batch, 1.0)
        eng._ws.fill_(255)
        eng.grad.fill_(float("nan"))
        P, S, loss = eng.train_fwd_bwd(batch, 1.0)
    finally:
        lib.gmc_set_fuse(prev)
    off = 0
    for (rp, cl, vl) in util.csrs_of(ds):
        f = CO.forward(rp, cl, vl, params["conv1.weight"], params["conv1.bias"], params["conv2.weight"], params["conv2.bias"])
        n = len(rp) - 1
        assert np.abs(P[off:off + n].cpu().numpy() - f["P"]).max() < PROB_TOL
        off += n
    ct = CO.CTrainer(params)
    ref_loss = ct.step(util.csrs_of(ds))
    # a row may decode differently from the oracle only on a float64 near-tie (then, and only then, the
    # gradients differ by design); everything else must hold
    Sk, Pk, off, near_ties = S.cpu().numpy(), P.cpu().numpy(), 0, 0
    for (rp, cl, vl), (_g, _a, nx_g, _t) in zip(util.csrs_of(ds), ds.values()):
        n = len(rp) - 1
        P64 = R.forward_dense_f64({k: torch.from_numpy(v) for k, v in params.items()}, nx_g)
        ref_s = P64.argmax(1)
        ref_s[:3] = [0, 1, 2]
        differ = np.nonzero(Sk[off:off + n] != ref_s)[0]
        srt = np.sort(P64[differ], axis=1)
        assert ((srt[:, 2] - srt[:, 1]) < 1e-6).all(), (differ, srt)
        near_ties += differ.size
        off += n
    if near_ties == 0:
        assert np.array_equal(loss.cpu().numpy(), ref_loss)
        ref = flat_ref_grads(ct)
        for k, g in eng.views(eng.grad).items():
            g, r = g.cpu().numpy().ravel(), ref[k]
            assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), k
    assert float(eng.grad[eng.count]) == float(loss.sum())


@pytest.mark.parametrize("fuse", [1, 0])
def test_relu_kink_is_the_only_gradient_mismatch(pkg, fuse):
    """(900, 7, 6), hidden 500: one of the 450,000 layer-1 pre-activations lands within fp32 rounding of 0;
    kernels and oracle may take different sides of the relu kink, and then ONE column of dW1 (and that
    entry of db1) differs.  Asserted instead of seeded away: every column whose gradient misses the 1e-4
    bar holds a pre-activation within fp32 noise of zero in the float64 formula; all other columns, dW2
    and db2 meet the bar."""
    specs = [(1000, 7, 5), (900, 7, 6), (500, 6, 8)]
    T, cfg, net, embed, opt, params = model_and_params(pkg, 500)
    ds = util.product_dataset(specs)
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    lib = pkg.hip.load()
    prev = lib.gmc_set_fuse(fuse)
    try:
        eng.train_fwd_bwd(batch, 1.0)
        eng._ws.fill_(255)
        eng.grad.fill_(float("nan"))
        P, S, loss = eng.train_fwd_bwd(batch, 1.0)
    finally:
        lib.gmc_set_fuse(prev)
    ct = CO.CTrainer(params)
    ref_loss = ct.step(util.csrs_of(ds))
    assert np.array_equal(loss.cpu().numpy(), ref_loss)
    # float64 pre-activations of layer 1: |y| < noise marks a kink column
    kink = np.zeros(500, bool)
    W1, b1 = params["conv1.weight"].astype(np.float64), params["conv1.bias"].astype(np.float64)
    for (_g, _a, nx_g, _t) in items:
        n = nx_g.number_of_nodes()
        a = np.zeros((n, n))
        for u, v in nx_g.edges():
            a[u, v] = a[v, u] = 1.0
        dis = 1.0 / np.sqrt(a.sum(1))
        t0 = dis[:, None] * (a @ W1[:n])                 # dinv o (A @ W1[:n])  (the X @ W1 row gather)
        pre = dis[:, None] * (a @ t0) + b1               # layer-1 pre-activation
        kink |= (np.abs(pre) < 1e-7).any(0)           # ~50 terms of ~3e-3: fp32 accumulation noise is ~1e-8
    ref = flat_ref_grads(ct)
    got = {k: g.cpu().numpy() for k, g in eng.views(eng.grad).items()}
    for k in ("conv2.weight", "conv2.bias"):   # (row f of dW2 = H[:, f]^T (...): H ~ 0 at the kink either way)
        r = ref[k]
        assert np.abs(got[k].ravel() - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), k
    tol1 = 1e-4 * max(1.0, np.abs(ref["conv1.weight"]).max())
    d1 = np.abs(got["conv1.weight"] - ref["conv1.weight"].reshape(1000, 500)).max(0)       # per column
    bad = np.nonzero(d1 > tol1)[0]
    assert set(bad) <= set(np.nonzero(kink)[0]), (bad, np.nonzero(kink)[0])
    tolb = 1e-4 * max(1.0, np.abs(ref["conv1.bias"]).max())
    badb = np.nonzero(np.abs(got["conv1.bias"] - ref["conv1.bias"]) > tolb)[0]
    assert set(badb) <= set(np.nonzero(kink)[0])
    assert kink.sum() <= 25 and len(bad) <= 2          # (a few of 500 columns qualify; at most the known one differs)


def test_adam_parity_step_by_step_on_the_reference_schedule(pkg):
    """One Adam step per graph (TrainingNeural.py:371-386), twelve steps, every step checked on its own:
    the oracle is re-synchronised to the device's (parameters, moments) before each step, so a wrong bias
    correction or moment update on a LATE step cannot hide behind accumulated trajectory noise.  Checked per
    step t: loss; m and v; and the parameter update on every entry whose gradient is at least 1 % of the
    largest (relative error of the update below 2 %: at t = 12 the bias correction is a factor 6.6)."""
    T, cfg, net, embed, opt, params = model_and_params(pkg, 16)
    specs = SPECS_SMALL * 3
    ds = util.product_dataset(specs)
    eng = net.engine()
    csrs = util.csrs_of(ds)
    ct = CO.CTrainer(params, lr=cfg.learning_rate)
    batches = [pkg.GraphBatch([it[0]], None, eng.device) for it in ds.values()]
    for t, (batch, csr) in enumerate(zip(batches, csrs), start=1):
        before = eng.flat[:eng.count].cpu().numpy().copy()
        ct.flat[:] = before
        ct.m[:] = eng.m[:eng.count].cpu().numpy()
        ct.v[:] = eng.v[:eng.count].cpu().numpy()
        ct.t = t - 1
        eng.sync_step_dev()
        _, _, loss = eng.train_step(batch, cfg.learning_rate, cfg.C)
        ref_loss = ct.step([csr])
        assert eng.step_count == t == ct.t
        if float(loss[0]) != float(ref_loss[0]):
            continue   # a near-tie decoded differently (covered elsewhere): the next step starts re-synchronised
        g = ct.grad
        big = np.abs(g) >= 1e-2 * np.abs(g).max()
        m, v = eng.m[:eng.count].cpu().numpy(), eng.v[:eng.count].cpu().numpy()
        assert np.abs(m - ct.m).max() <= 1e-4 * np.abs(ct.m).max()
        assert np.abs(v - ct.v).max() <= 2e-4 * np.abs(ct.v).max()
        upd = eng.flat[:eng.count].cpu().numpy() - before
        ref_upd = ct.flat - before
        rel = np.abs(upd - ref_upd)[big] / np.abs(ref_upd[big])
        assert big.sum() > 100 and rel.max() < 0.02, (t, rel.max())


def test_replayed_epoch_survives_a_larger_forward_and_follows_hyper_parameters(pkg):
    """hipGraph replay (one graph per epoch) must not be left pointing at freed scratch: epoch, then
    evaluate_model on a many-graph dataset (a far larger forward workspace), then epochs again == the same
    sequence with eager launches.  The replay must also follow a changed learning rate / C."""
    results = []
    for allow_graph in (True, False):
        T, cfg, net, embed, opt, params = model_and_params(pkg, 32)
        ds = util.product_dataset(SPECS_SMALL)
        big = util.product_dataset([(200, 7, 100 + i) for i in range(24)])
        tr = T._trainer_for(net, opt, cfg)
        tr.allow_graph = allow_graph
        out = [T.train_single_epoch(ds, net, opt, embed, cfg), T.train_single_epoch(ds, net, opt, embed, cfg)]
        ev = T.evaluate_model(net, big, cfg)
        junk = torch.full((1 << 22,), float("nan"), device="cuda")   # lands in whatever the allocator freed
        net.train()
        out.append(T.train_single_epoch(ds, net, opt, embed, cfg))
        opt.param_groups[0]["lr"] = 5e-3            # what optimizer.step() would use from now on
        cfg.C = 2.0
        out.append(T.train_single_epoch(ds, net, opt, embed, cfg))
        out.append(T.train_single_epoch(ds, net, opt, embed, cfg))
        torch.cuda.synchronize()
        del junk
        results.append((out, ev, {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}))
    (g_out, g_ev, g_sd), (e_out, e_ev, e_sd) = results
    assert g_out == e_out and g_ev == e_ev
    assert g_out[3] != g_out[2]                     # C = 2 doubles the loss scale: the replay saw it
    for k in g_sd:
        assert np.array_equal(g_sd[k], e_sd[k]), k


def test_empty_batch_clears_gradient_and_loss_tail(pkg):
    """A data-parallel rank whose shard of a step is empty: zero gradient AND zero loss in the tail slot
    (which still holds the previous step's batch loss)."""
    T, cfg, net, embed, opt, params = model_and_params(pkg, 16)
    ds = util.product_dataset(SPECS_SMALL[:2])
    eng = net.engine()
    batch = pkg.GraphBatch([it[0] for it in ds.values()], None, eng.device)
    eng.train_fwd_bwd(batch, 1.0)
    assert float(eng.grad[eng.count]) < 0
    empty = pkg.GraphBatch([], None, eng.device)
    eng.train_fwd_bwd(empty, 1.0)
    assert float(eng.grad[:eng.count + 1].abs().max()) == 0.0


def test_smallest_graphs_and_empty_inputs(pkg):
    """Edge sizes: a triangle whose three nodes are all terminals, 4- and 5-node graphs, an empty
    dataset (the reference returns 0 / average 0, TrainingNeural.py:371,566), a batch of zero graphs."""
    import networkx as nx
    from gcn_max_cut_amd.Training import TrainingNeural as T
    T_, cfg, net, embed, opt, params = model_and_params(pkg, 16)
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    from gcn_max_cut_amd import commons
    # the reference's terminal normalisation skips graphs with two or more of {0,1,2} among their
    # terminals (graphExtender.py:72-100), so 3- and 4-node graphs never pass it: build those items
    # by hand, the 5- and 6-node ones through it
    ds = {}
    for i, g in enumerate((nx.complete_graph(3), nx.cycle_graph(4))):
        nx.set_edge_attributes(g, 1, "weight")
        full = GE.extend_matrix_torch_2(commons.adjacency_tensor(g), 1000)
        ds[i] = [pkg.from_networkx(g), full, g, [0, 1, 2]]
    graphs = {0: nx.complete_graph(5), 1: nx.cycle_graph(6)}
    for g in graphs.values():
        nx.set_edge_attributes(g, 1, "weight")
    more = GE.process_graphs_from_folder(graphs, {0: [0, 3, 4], 1: [3, 4, 5]}, 1000)
    assert len(more) == 2
    assert len(GE.process_graphs_from_folder({0: nx.complete_graph(3)}, {0: [0, 1, 2]}, 1000)) == 0   # skipped
    for it in more.values():
        ds[len(ds)] = it
    eng = net.engine()
    items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = eng.train_fwd_bwd(batch, 1.0)
    ct = CO.CTrainer(params)
    ref_loss = ct.step(util.csrs_of(ds))
    assert np.array_equal(loss.cpu().numpy(), ref_loss)
    assert float(loss[0]) == -3.0   # triangle, three different terminals: every edge is cut
    ref = flat_ref_grads(ct)
    for k, g in eng.views(eng.grad).items():
        assert np.abs(g.cpu().numpy().ravel() - ref[k]).max() <= 1e-4 * max(1.0, np.abs(ref[k]).max()), k
    assert T.evaluate_model(net, {}, cfg) == {'average_loss': 0, 'total_loss': 0.0, 'num_samples': 0}
    assert T.train_single_epoch({}, net, opt, embed, cfg) == 0.0
    empty = pkg.GraphBatch([], None, eng.device)
    Pe, Se, le = eng.forward(empty, 1.0, want_loss=True)
    assert Pe.shape[0] == 0 and le.numel() == 0


def test_checkpoint_round_trip(pkg, tmp_path, monkeypatch):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    monkeypatch.chdir(tmp_path)
    ds = util.product_dataset(SPECS_SMALL)
    net, best, epoch, emb, hist = T.train_model(ds, T.TrainingConfig(
        n_nodes=1000, hidden_dim=16, number_epochs=3, save_directory="m.pth", save_frequency=1))
    assert len(hist) == 3 and best == min(hist)
    ck = torch.load(tmp_path / "final_m.pth", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "loss_history", "inputs", "config"}
    assert sorted(ck["optimizer"]["state"].keys()) == [0, 1, 2, 3]
    assert set(ck["optimizer"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    net2, inputs, cfg2 = T.load_neural_model(str(tmp_path / "final_m.pth"), T.TrainingConfig(n_nodes=1000, hidden_dim=16))
    assert tuple(inputs.shape) == (1000, 1000) and cfg2.hidden_dim == 16
    a, b = T.evaluate_model(net, ds, cfg2), T.evaluate_model(net2, ds, cfg2)
    assert a == b


def test_decode_and_post_processing_match_reference_goldens(pkg):
    """gmc_decode_sample_f32 against vectors recorded from the reference's own
    TestingNeuralNetwork functions (tests/golden/decode.json) - exact."""
    import json, os
    from gcn_max_cut_amd.Testing import TestingNeuralNetwork as TN
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "decode.json")))["cases"]
    for case in cases:
        g = R.regular_graph(case["n"], case["d"], case["seed"])
        probs = torch.tensor(case["probs"], dtype=torch.float32)
        assert TN.simple_partition_assignment(probs) == case["simple_assignment"]
        assert TN.calculate_cut_value(case["simple_assignment"], g) == case["simple_cut"]
        np.random.seed(case["np_seed"])
        best, cut = TN.post_processing_optimization(probs, g, iterations=case["post_iterations"])
        assert cut == case["post_cut"] and best == case["post_assignment"]
        # first two samples individually (iterations=1 consumes exactly one sample's draws)
        np.random.seed(case["np_seed"])
        s0, c0 = TN.post_processing_optimization(probs, g, iterations=1)
        s1, _ = TN.post_processing_optimization(probs, g, iterations=1)
        assert s0 == case["sample0"] and s1 == case["sample1"] and c0 == case["sample0_cut"]


def test_sampler_boundary_draws_follow_the_pinned_numpy_semantics(pkg, monkeypatch):
    """Testing/TestingNeuralNetwork.py:33-39 - `cumulative_prob = 0; cumulative_prob += np.float32; rand_val <
    cumulative_prob` means different arithmetic under different NumPy versions.  decode.hip, the host form and
    the oracle follow the reference's PINNED NumPy 1.x (envList.txt:105): float64 running sum, float64 compare.
    Directed draws one ulp either side of the cumulative boundaries (where a float32 sum / float32 compare, i.e.
    NumPy >= 2, decides differently): kernel == host form == oracle, and != the NumPy-2 reading."""
    from gcn_max_cut_amd.Testing import TestingNeuralNetwork as TN
    n = 43
    g = R.regular_graph(n + 1, 5, 3)
    n = g.number_of_nodes()
    rng = np.random.RandomState(1)
    probs = np.empty((n, 3), np.float32)
    draws = np.empty(n - 3, np.float64)
    kinds = 0
    for l in range(n):
        a, b = rng.uniform(0.05, 0.45, 2)
        p0, p1 = np.float32(a), np.float32(b)
        probs[l] = [p0, p1, np.float32(1.0) - p0 - p1]
        if l < 3:
            continue
        c0 = float(p0)
        c1_64 = float(p0) + float(p1)
        c1_32 = float(np.float32(p0 + p1))
        kind = (l - 3) % 6
        if kind == 0:
            r = np.nextafter(c0, 0.0)                  # double: r < c0 -> class 0; float32(r) == c0 -> NumPy 2 says class >= 1
        elif kind == 1:
            r = c0                                     # not < c0 either way
        elif kind == 2:
            r = np.nextafter(c1_64, 0.0)               # just below the double sum
        elif kind == 3:
            r = c1_64
        elif kind == 4:
            r = 0.5 * (c1_64 + c1_32)                  # between the double and the float32 sum (when they differ)
        else:
            r = np.nextafter(c0, 1.0)
        kinds |= 1 << kind
        draws[l - 3] = r
    assert kinds == 63

    class Fake:   # np.random.rand: array form (the GPU path) and scalar form (host / oracle), same stream
        def __init__(self):
            self.i = 0

        def __call__(self, *shape):
            if shape:
                assert shape == (1, n - 3)
                return draws.reshape(1, -1).copy()
            v = draws[self.i]
            self.i += 1
            return float(v)

    monkeypatch.setattr(np.random, "rand", Fake())
    got, _cut = TN.post_processing_optimization(torch.from_numpy(probs), g, iterations=1)      # gmc_decode_sample_f32
    monkeypatch.setattr(np.random, "rand", Fake())
    host = TN.assign_partitions(probs)
    ref = R.sample_partition(probs, Fake())
    other = R.sample_partition_numpy2(probs, Fake())
    assert list(got) == ref == host
    assert ref != other and sum(a != b for a, b in zip(ref, other)) >= 5     # the draws really sit on the fault line


def test_inference_harness_matches_oracle_post_processing(pkg):
    """test_multiple_graphs (config 5 shape: mixed sizes) == oracle decode with the same RNG."""
    from gcn_max_cut_amd.Testing import TestingNeuralNetwork as TN
    T, cfg, net, *_rest, params = model_and_params(pkg, 64)
    specs = [(50, 6, 50001), (100, 7, 100001), (200, 8, 200001), (300, 6, 300001), (500, 7, 500001)]
    ds = util.product_dataset(specs)
    np.random.seed(0)
    results, by_size = TN.test_multiple_graphs(net, ds, [50, 100, 200, 300, 500], post_processing_iterations=20, verbose=False)
    assert len(results) == 5 and all(r["success"] for r in results)
    np.random.seed(0)
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    for r, (g, a_pad, nx_g, _t) in zip(results, ds.values()):
        P = torch.from_numpy(r["node_probabilities"])
        assert r["simple_assignment"] == R.partition_of(P).tolist()
        assert r["simple_cut"] == R.cut_value(r["simple_assignment"], nx_g)
        best, cut = R.post_process(r["node_probabilities"], nx_g, 20, np.random.rand)
        assert r["post_cut"] == cut and r["post_assignment"] == best
        assert r["improvement"] == cut - r["simple_cut"]
    assert by_size[500]["post_processed"]["cut_values"] == [results[4]["post_cut"]]
    # batched throughput form gives the same numbers
    np.random.seed(0)
    fast = TN.decode_dataset(net, ds, 20)
    for r, f in zip(results, fast):
        assert (f["simple_cut"], f["post_cut"], f["post_assignment"]) == (r["simple_cut"], r["post_cut"], r["post_assignment"])


def test_forward_with_arbitrary_dense_features(pkg):
    """net(g, X) with X that is not the adjacency: dense layer-1 GEMM + HIP kernels (inference)."""
    T, cfg, net, *_rest, params = model_and_params(pkg, 32)
    ds = util.product_dataset([(60, 5, 77)])
    (g, a_pad, nx_g, _t), = ds.values()
    torch.manual_seed(1)
    X = torch.randn(60, 1000)
    net.eval()
    with torch.no_grad():
        P = net(g, X.cuda()).cpu()
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    ref = R.forward(tp, R.graph_from_networkx(nx_g), X)
    assert float((P - ref).abs().max()) < PROB_TOL
    net.train()
    with pytest.raises(NotImplementedError):
        net(g, X.cuda())


def test_dropout_training_path(pkg):
    """F.dropout(h, p, training) between the layers (TrainingNeural.py:82; TrainingConfig.dropout :43).  Every
    reference configuration uses p = 0; for p > 0 the mask comes from the library's counter-based hash, so the
    random stream is NOT torch's (parity unpinned for the mask).  What is checked instead:
    eval mode ignores p; the same torch seed gives the same mask, the next call a different one; kept units are
    scaled by 1/(1-p) and the keep rate is 1-p; the gradient through the autograd path equals central finite
    differences with the mask held fixed; the fused training entry (gmc_train_fwd_bwd) and the autograd path
    (gmc_forward + gmc_backward_from_gp) give the same gradient for the same seed; an epoch of the trainer runs."""
    from gcn_max_cut_amd.Training import TrainingNeural as T
    p_drop, hidden = 0.3, 64
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=hidden, dropout=p_drop)
    torch.manual_seed(0)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    params = util.np_params(net.state_dict())
    ds = util.product_dataset([(80, 7, 1), (60, 6, 2)])
    (g, a_pad, nx_g, _t) = ds[0]
    tp = {k: torch.from_numpy(v) for k, v in params.items()}

    net.eval()
    with torch.no_grad():
        P_eval = net(g, a_pad).cpu()
    assert float((P_eval - R.forward(tp, R.graph_from_networkx(nx_g), a_pad.cpu())).abs().max()) < PROB_TOL

    net.train()
    with torch.no_grad():
        torch.manual_seed(5); P1 = net(g, a_pad).cpu()
        torch.manual_seed(5); P2 = net(g, a_pad).cpu()
        P3 = net(g, a_pad).cpu()
    assert torch.equal(P1, P2) and not torch.equal(P1, P3) and not torch.equal(P1, P_eval)

    # keep rate and 1/(1-p) scale: W1 = 0, b1 = 1 -> relu = 1 everywhere, H_dropped = mask / (1-p);
    # W2[:, 0] = c, rest 0, b2 = 0 -> Z[r,0] - Z[r,1] = log(P0/P1) = c * mean over the neighbours of sum_f H_dropped
    eng = net.engine()
    with torch.no_grad():
        saved = eng.flat.clone()
        v = eng.views()
        v["conv1.weight"].zero_(); v["conv1.bias"].fill_(1.0); v["conv2.weight"].zero_(); v["conv2.weight"][:, 0] = 0.01
        v["conv2.bias"].zero_()
        torch.manual_seed(11)
        P = net(g, a_pad).cpu().double()
        z = torch.log(P[:, 0] / P[:, 1])
        assert abs(float(z.mean()) / (0.01 * hidden) - 1.0) < 0.03        # E[kept / (1-p)] = F
        assert float(z.std()) > 1e-3                                        # ... and it IS random
        net.eval()
        Pe = net(g, a_pad).cpu().double()
        assert torch.allclose(torch.log(Pe[:, 0] / Pe[:, 1]), torch.full((g.number_of_nodes(),), 0.01 * hidden, dtype=torch.float64), atol=1e-4)
        net.train()
        eng.flat.copy_(saved)

    # gradient through the autograd path vs central differences, mask held fixed by the seed
    Wt = torch.from_numpy(np.random.RandomState(3).standard_normal((g.number_of_nodes(), 3)).astype(np.float32)).cuda()

    def loss_at(seed):
        torch.manual_seed(seed)
        return (net(g, a_pad) * Wt).sum()

    net.zero_grad()
    loss_at(21).backward()
    named = dict(net.named_parameters())
    for name, idx in (("conv2.weight", (5, 1)), ("conv2.bias", (2,)), ("conv1.bias", (7,)), ("conv1.weight", (3, 9))):
        prm = named[name]
        gval = float(prm.grad[idx])
        eps = 2e-2
        with torch.no_grad():
            old = float(prm[idx])
            prm[idx] = old + eps; lp = float(loss_at(21))
            prm[idx] = old - eps; lm = float(loss_at(21))
            prm[idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - gval) <= 0.05 * max(abs(gval), abs(fd)) + 2e-3, (name, fd, gval)

    # the same seed through both gradient routes: fused training entry vs forward + backward_from_gp
    batch = pkg.GraphBatch([g], None, eng.device)
    torch.manual_seed(33)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    eng.set_dropout(p_drop, seed)
    eng.train_fwd_bwd(batch, 1.0)
    eng.set_dropout(0.0)
    fused = {k: gr.clone() for k, gr in eng.views(eng.grad).items()}
    net.zero_grad()
    torch.manual_seed(33)
    Pt = net(g, a_pad)
    T.compute_loss(T.apply_max_to_one_hot(T.override_fixed_nodes(Pt)), a_pad.cuda(), 0.0, 1.0, 1000.0).backward()
    for k, prm in named.items():
        r = fused[k]
        assert float((prm.grad - r).abs().max()) <= 1e-4 * max(1.0, float(r.abs().max())), k

    # and an epoch of the trainer (eager steps, a fresh mask per step); eval afterwards is dropout-free
    before = eng.flat.clone()
    loss = T.train_single_epoch(ds, net, opt, embed, cfg)
    assert np.isfinite(loss) and loss < 0 and not torch.equal(before, eng.flat)
    assert eng.dropout_state()[0] == 0.0
    out = T.evaluate_model(net, ds, cfg)
    tp2 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref_total = sum(float(R.graph_loss(tp2, R.graph_from_networkx(it[2]), it[1].cpu(), 1.0)[0]) for it in ds.values())
    assert out["total_loss"] == ref_total


def test_bench_line_contract(pkg):
    """bench.py prints ONE JSON line with the keys the driver reads (small run: 8 graphs per GPU)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--graphs-per-gpu", "8", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # `roofline` = the dominant kernel of the TIMED step (fused layer-1 backward or forward) on the bytes it must move,
    # with the LDS line beside the HBM one; the stand-alone SpMM the metric names is `roofline_spmm`
    r = d["roofline"]
    assert r["kernel"] in ("fwd1_lds_kernel", "bwd1_reg_kernel") and r["in_timed_region"] is True
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["limited_by"] == "lds"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0 and len(r["also"]) == 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["mean_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["lds"]["peak"] == 256 * 256 * 2.4 and 0 < r["lds"]["frac"] < 1
    rs = d["roofline_spmm"]
    assert rs["bound"] == "hbm" and rs["peak"] == 8000.0 and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-9
    assert "roofline_step" not in d and "survey_unfused_GBps" not in json.dumps(d)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] * 160 / (1e3 * 8) - 1) < 1e-6
    assert d["parity"]["argmax_equal"] and d["parity"]["max_abs_prob_diff"] <= 1e-4
    assert d["parity"]["max_abs_prob_diff_one_kernel_per_op"] <= 1e-4
    q = d["sequential"]
    assert q["optimizer_steps_per_epoch"] == 8 and q["value"] > 0
    assert {"A_all_threads", "A_1_thread", "B_sparse_c_1_thread"} <= set(c["variants"])
    assert c["variants"]["A_1_thread"]["cores"] == 1 and c["cpu_model"]
    assert c["gpu_sequential_over_cpu"] > 0 and c["gpu_sequential_over_cpu_variant_B"] > 0
    # other degrees / graph families on the same fused kernels, each with its parity gate
    ow = d["other_workloads"]
    assert set(ow) == {"d8", "d12", "gnp", "c1_20x500_d6to8"}
    assert ow["c1_20x500_d6to8"]["table_slots"] == 8 and ow["c1_20x500_d6to8"]["max_degree"] == 8
    assert (ow["d8"]["table_slots"], ow["d8"]["live_slots"]) == (8, 8) and ow["d8"]["overflow_blocks"] == 0
    assert (ow["d12"]["table_slots"], ow["d12"]["live_slots"]) == (16, 12) and ow["d12"]["overflow_blocks"] == 0
    assert ow["gnp"]["table_slots"] == 16 and ow["gnp"]["overflow_blocks"] > 0 and ow["gnp"]["max_degree"] > 16
    for w in ow.values():
        assert w["ms_per_step"] > 0 and w["parity"]["argmax_equal"] and w["parity"]["max_abs_prob_diff"] <= 1e-4
        assert {"fwd1_fused", "head", "bwd1_fused", "finish"} == set(w["kernels_us"])
    # the data-parallel step on one rank over RCCL, eager against hipGraphs
    dpp = d["dp_shard_profile"]
    assert dpp["backend"] == "nccl" and list(dpp["by_shard"]) == ["8"]
    rec = dpp["by_shard"]["8"]
    assert rec["eager"]["ms_per_step"] > 0 and rec["graphs"]["ms_per_step"] > 0 and rec["allreduce_single_rank_us"] > 0
    assert rec["faster"] in ("eager", "graphs") and rec["eager"]["host_enqueue_us"] > 0


def test_bench_two_rank_rehearsal_on_one_gpu(pkg):
    """`bench.py --gpus 2` launches its own two ranks (children, before any GPU call); on a one-GPU box
    GCN_MAXCUT_BENCH_REHEARSE=1 puts both on device 0 over gloo: the N > 1 control flow end to end -
    sharded datasets, replica sync, hipGraphs around the eager all-reduce, both scalings in one line."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GCN_MAXCUT_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--graphs-per-gpu", "6"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["launch"].startswith("eager launches around the all-reduce")
    assert d["strong_scaling"]["graphs_per_gpu"] == 6 and d["strong_scaling"]["value"] > 0
    assert d["allreduce_ms_per_step"] > 0 and d["value"] > 0
    assert "cpu_baseline" not in d


@pytest.mark.timeout(900)
def test_bench_single_rank_over_rccl(pkg):
    """RCCL itself on the one GPU a test box has: GCN_MAXCUT_DP_SINGLE_RANK=1 makes a world of ONE rank run the
    data-parallel step sequence over backend "nccl" - communicator set-up, replica broadcast, the [gradient | loss]
    all-reduce on the launch stream between eager launches (the default; GCN_MAXCUT_DP_GRAPHS=1 replays hipGraphs there).
    The loss must be the one the single-GPU fused step reaches after the same number of steps."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        base.pop(k, None)
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
            "--graphs-per-gpu", "6", "--no-cpu-baseline", "--no-sequential", "--no-probe"]
    lines = {}
    for tag, env in (("rccl", dict(base, GCN_MAXCUT_DP_SINGLE_RANK="1")), ("single", base)):
        out = subprocess.run(args, capture_output=True, text=True, timeout=400, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        (line,) = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
        lines[tag] = json.loads(line)
    d, s = lines["rccl"], lines["single"]
    assert d["backend"] == "nccl" and d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1"
    assert d["launch"].startswith("eager launches around the all-reduce") and s["launch"].startswith("eager launches queued")
    assert d["allreduce_ms_per_step"] > 0 and d["value"] > 0
    assert d["last_loss"] == s["last_loss"]          # same six graphs, same six steps, integer-valued loss


def _slab_of(W1: np.ndarray) -> np.ndarray:
    """gmc_model.W1_slab: [ceil(F/16)][N][16], pad columns zero."""
    N, F = W1.shape
    pad = np.zeros((N, (F + 15) // 16 * 16), np.float32)
    pad[:, :F] = W1
    return pad.reshape(N, -1, 16).transpose(1, 0, 2).copy()


@pytest.mark.parametrize("hidden,gps", [(500, 2), (72, 2), (72, 1)])      # 72 = 4.5 slabs: a partly padded last slab;
def test_w1_slab_copy_layout_and_bitwise_identical_training(pkg, hidden, gps, monkeypatch):
    """The slab copy of conv1.weight (gmc_model.W1_slab): gmc_w1_slab_f32 lays it out as documented; a trainer
    whose fused forward reads it and whose Adam sweep keeps it current ends up with BIT-identical parameters,
    losses and copy (the copy changes where the W1 tile is fetched from, not what is computed) - on one GPU
    (whole-step hipGraph) and on the data-parallel step sequence (graphs either side of the all-reduce)."""
    specs = [(1000, 7, 61), (700, 7, 62), (1000, 7, 63), (520, 6, 64)]
    ds = util.product_dataset(specs)
    monkeypatch.setenv("GCN_MAXCUT_DP_GRAPHS", "1")   # the data-parallel sequence replayed from hipGraphs (opt-in)

    class WithoutFusedStep:           # an engine that offers only what the N > 1 branch uses
        def __init__(self, eng):
            self._eng = eng

        def __getattr__(self, name):
            if name == "train_step":
                raise AttributeError(name)
            return getattr(self._eng, name)

    for dp_sequence in (False, True):
        runs = []
        for slab in (True, False):
            T, cfg, net, embed, opt, params = model_and_params(pkg, hidden, seed=9)
            eng = net.engine()
            eng.slab_enabled = slab and eng.slab_enabled
            tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=gps, engine=WithoutFusedStep(eng) if dp_sequence else eng)
            if dp_sequence:
                tr.dp = True                          # the all-reduce itself is a no-op without a process group
            losses = [tr.epoch(ds) for _ in range(3)]
            torch.cuda.synchronize()
            if slab:
                assert eng.w1_slab is not None and (tr._graph is not None or tr._dp_graph is not None)
                got = eng.w1_slab.cpu().numpy().reshape(-1, 1000, 16)
                assert np.array_equal(got, _slab_of(eng.views()["conv1.weight"].cpu().numpy()))   # kept current
            runs.append((losses, eng.flat.cpu().numpy().copy()))
        assert runs[0][0] == runs[1][0]
        assert np.array_equal(runs[0][1], runs[1][1])


def test_w1_slab_copy_follows_torch_writes(pkg):
    """Anything torch writes into the parameters between two steps of the fused trainer (optimizer surgery,
    load_state_dict, an in-place edit) must reach the next forward: the engine re-builds the slab copy when a
    version counter of the flat buffer or of an adopted parameter moved."""
    ds = util.product_dataset([(1000, 7, 71), (640, 7, 72)])
    runs = []
    for slab in (True, False):
        T, cfg, net, embed, opt, params = model_and_params(pkg, 64, seed=3)
        eng = net.engine()
        eng.slab_enabled = slab and eng.slab_enabled
        tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=2)
        out = [tr.epoch(ds), tr.epoch(ds)]
        with torch.no_grad():
            net.conv1.weight.mul_(0.5)                               # through the nn.Parameter
        out.append(tr.epoch(ds))
        eng.views()["conv1.weight"][:300].add_(0.01)                 # through a view of the flat buffer
        out.append(tr.epoch(ds))
        state = {k: v.clone() for k, v in net.state_dict().items()}
        out.append(tr.epoch(ds))
        net.load_state_dict(state)                                   # roll one step back
        out.append(tr.epoch(ds))
        torch.cuda.synchronize()
        assert out[5] == out[4]                                      # the same step from the same weights
        runs.append((out, eng.flat.cpu().numpy().copy()))
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])


def test_slab_and_host_memory_entry_points_directly(pkg):
    """The C ABI of the round-2 additions, called directly: gmc_w1_slab_f32 (layout), gmc_adam_devstep_model_f32
    (== gmc_adam_devstep_f32 on the parameters, and the slab copy follows), gmc_host_device_pointer +
    gmc_publish_f32 (device floats land in pinned host memory the host can read after the stream is idle)."""
    import ctypes as C
    lib, hip = pkg.hip.load(), pkg.hip
    N, F = 1000, 72
    count = N * F + F + 3 * F + 3
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(count, generator=g).cuda()
    grad = (0.1 * torch.randn(count, generator=g)).cuda()
    slab = torch.empty(int(lib.gmc_w1_slab_floats(N, F)), device="cuda")
    st = hip.stream()
    assert lib.gmc_w1_slab_f32(hip.ptr(p0), N, F, hip.ptr(slab), st) == 0
    assert np.array_equal(slab.cpu().numpy().reshape(-1, N, 16), _slab_of(p0[:N * F].view(N, F).cpu().numpy()))
    runs = []
    for with_slab in (False, True):
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        for _ in range(3):
            if with_slab:
                rc = lib.gmc_adam_devstep_model_f32(hip.ptr(p), hip.ptr(grad), hip.ptr(m), hip.ptr(v), N, F, hip.ptr(slab),
                                                    1e-3, 0.9, 0.999, 1e-8, hip.ptr(step), st)
            else:
                rc = lib.gmc_adam_devstep_f32(hip.ptr(p), hip.ptr(grad), hip.ptr(m), hip.ptr(v), count, 1e-3, 0.9, 0.999,
                                              1e-8, hip.ptr(step), st)
            assert rc == 0
        assert int(step.item()) == 3
        runs.append(p.cpu().numpy())
    assert np.array_equal(runs[0], runs[1])
    assert np.array_equal(slab.cpu().numpy().reshape(-1, N, 16), _slab_of(runs[1][:N * F].reshape(N, F)))
    host = torch.full((8,), float("nan")).pin_memory()
    dev = hip.mapped_ptr(host)
    assert dev, "pinned host memory must be mapped into the device"
    src = torch.arange(8, dtype=torch.float32, device="cuda") - 3.5
    assert lib.gmc_publish_f32(hip.ptr(src), 8, dev, st) == 0
    torch.cuda.current_stream().synchronize()
    assert np.array_equal(host.numpy(), src.cpu().numpy())
    assert hip.mapped_ptr(torch.zeros(4)) is None      # pageable memory: refused, the trainer then copies instead


def test_one_step_epochs_same_result_on_every_launch_path(pkg, monkeypatch):
    """One batched step per epoch - the bench workload's shape - through the three launch paths: eager launches
    with the losses stored straight into pinned host memory (default), a graph replay per step, and plain eager
    launches with a device loss buffer, a copy and a stream synchronisation.  Same losses, bit-identical models."""
    specs = [(1000, 7, 91), (640, 7, 92), (1000, 7, 93)]
    ds = util.product_dataset(specs)
    runs = {}
    for path, env in (("direct", {}), ("graph", {"GCN_MAXCUT_EAGER_SINGLE_STEP": "0"}),
                      ("copy", {"GCN_MAXCUT_LOSS_ZEROCOPY": "0", "GCN_MAXCUT_POLL_LOSS": "0", "GCN_MAXCUT_HIPGRAPH": "0"})):
        for k in ("GCN_MAXCUT_EAGER_SINGLE_STEP", "GCN_MAXCUT_LOSS_ZEROCOPY", "GCN_MAXCUT_POLL_LOSS", "GCN_MAXCUT_HIPGRAPH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        T, cfg, net, embed, opt, params = model_and_params(pkg, 128, seed=17)
        tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=len(specs))
        losses = [tr.epoch(ds) for _ in range(6)]
        torch.cuda.synchronize()
        if path == "direct":
            assert tr._graph is None and tr._loss_host_dev is not None
        elif path == "graph":
            assert tr._graph is not None
        else:
            assert tr._graph is None and tr._loss_host_dev is None
        runs[path] = (losses, net.engine().flat.cpu().numpy().copy())
    for path in ("graph", "copy"):
        assert runs[path][0] == runs["direct"][0], path
        assert np.array_equal(runs[path][1], runs["direct"][1]), path


def test_polled_losses_nan_is_a_value_and_first_epoch_is_fast(pkg):
    """The host watches the pinned loss slots for a BIT PATTERN ("not landed"), not for NaN: a loss that genuinely is
    NaN (here: C = NaN) comes back at once instead of costing the 5 s polling deadline per step, and an ordinary
    polled epoch finishes far below the deadline without ever taking the fallback."""
    import math, time as _time
    ds = util.product_dataset([(300, 7, 91), (200, 6, 92), (260, 8, 93)])
    for gps in (3, 1):       # batched step (eager, zero-copy stores) and the reference schedule (hipGraph per epoch)
        T, cfg, net, embed, opt, _ = model_and_params(pkg, 64)   # (a NaN step ruins the weights: fresh model each time)
        tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=gps)
        t0 = _time.perf_counter()
        first = tr.epoch(ds)
        assert _time.perf_counter() - t0 < 2.0 and first < 0 and tr.deadline_hits == 0 and tr._poll
        cfg_nan = T.TrainingConfig(n_nodes=1000, hidden_dim=64, C=float("nan"))
        tr.config = cfg_nan
        for _ in range(3):   # (the first epoch with a new C re-captures the graph)
            t0 = _time.perf_counter()
            got = tr.epoch(ds)
            assert math.isnan(got)
        assert _time.perf_counter() - t0 < 1.0 and tr.deadline_hits == 0 and tr._poll
        tr.config = cfg
    # a middle item replaced in a large dict is invisible to prepare()'s probes: invalidate() re-plans
    T, cfg, net, embed, opt, _ = model_and_params(pkg, 64)
    big = util.product_dataset([(60, 5, 200 + i) for i in range(40)])
    tr = T.FusedTrainer(net, opt, cfg, graphs_per_step=40)
    tr.epoch(big)
    plan = tr._batches
    keys = list(big)
    other = util.product_dataset([(64, 6, 999)])
    big[keys[3]] = other[0]                       # not one of the probed positions (0, 5, 11, ..., 39)
    tr.epoch(big)
    assert tr._batches is plan
    tr.invalidate()
    tr.epoch(big)
    assert tr._batches is not plan and tr._batches[0].R == 39 * 60 + 64
    big[keys[0]] = other[0]                       # a probed position: seen without help
    tr.epoch(big)
    assert tr._batches[0].R == 38 * 60 + 2 * 64
