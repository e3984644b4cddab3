"""Host-side logic of the drop-in API that needs no GPU: configuration, GraphExtender,
graph handle and batch layout, checkpoints, training-loop bookkeeping, import roots."""
import copy
import os
import pickle

import networkx as nx
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R
from tests import util


@pytest.fixture(scope="module")
def T(built):
    from gcn_max_cut_amd.Training import TrainingNeural
    return TrainingNeural


def test_training_config_defaults_and_derived_fields(T):
    c = T.TrainingConfig()
    assert (c.n_nodes, c.dim_embedding, c.hidden_dim, c.number_classes) == (1000, 1000, 500, 3)
    assert (c.learning_rate, c.number_epochs, c.tolerance, c.patience) == (0.001, 1000, 1e-4, 20)
    assert (c.prob_threshold, c.A, c.C, c.penalty, c.save_directory, c.save_frequency) == (0.5, 0.0, 1.0, 1000.0, None, 100)
    c = T.TrainingConfig(n_nodes=4096)
    assert c.dim_embedding == 4096 and c.hidden_dim == 2048
    assert [f for f in c.__dataclass_fields__] == [
        "n_nodes", "dim_embedding", "hidden_dim", "dropout", "number_classes", "learning_rate", "number_epochs",
        "tolerance", "patience", "prob_threshold", "A", "C", "penalty", "save_directory", "save_frequency"]


def test_model_layout_matches_dgl_graphconv(T):
    net, embed, opt = T.setup_model_and_optimizer(T.TrainingConfig())
    sd = net.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "conv1.weight": (1000, 500), "conv1.bias": (500,), "conv2.weight": (500, 3), "conv2.bias": (3,)}
    assert sum(p.numel() for p in net.parameters()) == 502003
    assert float(sd["conv1.bias"].abs().max()) == 0.0
    assert float(sd["conv1.weight"].abs().max()) <= (6 / 1500) ** 0.5 + 1e-7   # xavier-uniform bound
    assert tuple(embed.weight.shape) == (1000, 1000)
    assert len(opt.param_groups[0]["params"]) == 5       # the unused embedding is in the optimizer (Q1)


def test_graph_extender_matches_oracle_and_keeps_reference_quirks(built, tmp_path, monkeypatch):
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    monkeypatch.chdir(tmp_path)
    specs = [(40, 5, 11), (30, 4, 12), (36, 6, 13)]
    graphs, terms = util.make_graphs(specs)
    terms = {0: [5, 9, 11], 1: [2, 7, 9], 2: [0, 1, 5]}          # case 1, case 2, skipped
    g2, t2 = copy.deepcopy(graphs), copy.deepcopy(terms)
    ds = GE.process_graphs_from_folder(graphs, terms, 1000)
    ods = R.make_dataset(g2, t2, 1000)
    assert list(ds) == list(ods) == [0, 1]
    for k in ds:
        h, a_pad, nx_g, t = ds[k]
        og, oa, onx, ot = ods[k]
        assert t == ot == [0, 1, 2] and a_pad.dtype == torch.float32
        assert torch.equal(a_pad.cpu(), oa)
        assert h.number_of_nodes() == og.n and h.number_of_edges() == og.number_of_edges() == 2 * nx_g.number_of_edges()
        assert sorted(map(sorted, nx_g.edges())) == sorted(map(sorted, onx.edges()))
        rp, cl, _ = CO.csr_of(nx_g)
        assert np.array_equal(h.rowptr, rp) and np.array_equal(h.col, cl)
    assert graphs[0] is ds[0][2]                      # caller's graph relabelled in place (Q11)
    assert terms[1] == [2, 7, 9] and terms[1] is not None
    # batch flush resets the returned dict (Q11)
    graphs, terms = util.make_graphs(specs)
    out = GE.process_graphs_from_folder(graphs, {0: [5, 9, 11], 1: [6, 7, 9], 2: [4, 8, 9]}, 1000,
                                        save_batch_size=2, output_filename_prefix="flush")
    assert list(out) == [2] and os.path.exists("flush_2.pkl")
    assert list(pickle.load(open("flush_2.pkl", "rb"))) == [0, 1]
    with pytest.raises(ValueError):
        GE.extend_matrix_torch_2(torch.zeros(5, 5), 3)


def test_graph_handle_pickles_and_answers_like_a_dgl_graph(built):
    g = R.regular_graph(50, 6, 3)
    h = built.from_networkx(g).to("cpu")
    h2 = pickle.loads(pickle.dumps(h))
    assert h2.number_of_nodes() == 50 and h2.number_of_edges() == 300
    assert np.array_equal(h2.col, h.col) and int(h.in_degrees().sum()) == 300
    src, dst = h.edges()
    assert sorted(zip(src.tolist(), dst.tolist())) == sorted([(u, v) for u, v in g.edges()] + [(v, u) for u, v in g.edges()])
    # weighted graph: edge values follow the adjacency the model is called with
    for u, v in g.edges():
        g[u][v]["weight"] = 1 + (u + v) % 3
    hw = built.from_networkx(g)
    a = R.dense_adjacency(g, 1000)
    vals = hw.edge_values(a)
    assert vals is not None and np.array_equal(vals, hw.weight)
    with pytest.raises(NotImplementedError):
        bad = a.clone(); bad[0, 999] = 1.0
        hw.edge_values(bad)
    iso = nx.path_graph(4); iso.add_node(4)
    with pytest.raises(built.DGLError):
        built.from_networkx(iso).check_degrees()


def test_batch_layout_csr_and_ell(built):
    gs = [R.regular_graph(n, d, s) for n, d, s in [(30, 5, 1), (40, 8, 2), (24, 3, 3)]]
    hs = [built.from_networkx(g) for g in gs]
    b = built.BatchArrays(hs)
    assert (b.B, b.R, b.nnz, b.n_max, b.uniform_n, b.ell_width) == (3, 94, 30 * 5 + 40 * 8 + 24 * 3, 40, 0, 8)
    assert b.goff.tolist() == [0, 30, 70, 94]
    for g, h in enumerate(hs):
        r0 = int(b.goff[g])
        for l in range(h.n):
            e0, e1 = b.rowptr[r0 + l], b.rowptr[r0 + l + 1]
            nb = h.col[h.rowptr[l]:h.rowptr[l + 1]]
            assert np.array_equal(b.lcol[e0:e1], nb) and np.array_equal(b.gcol[e0:e1], nb + r0)
            slots = b.ell[r0 + l].tolist()     # neighbours in the bank-aware slot order + padding
            assert sorted(x for x in slots if x < h.n) == sorted(nb.tolist())
            assert all(h.n <= x < h.n + 4 for x in slots if x >= h.n) and len(slots) == 8
            assert b.dinv[r0 + l] == np.float32(1.0) / np.sqrt(np.float32(len(nb)))
    assert b.vals is None and b.ell_vals is None
    assert built.BatchArrays([hs[0], hs[0]]).uniform_n == 30
    big = built.from_networkx(nx.random_regular_graph(20, 64, seed=1))
    assert built.BatchArrays([big]).ell is None                     # degree > 16: row kernels only


def test_batch_layout_table_width_and_overflow_lists(built):
    """Rows longer than the table: the first W neighbours (CSR order) sit in the ELL row, the rest in blocks of
    eight overflow ids (padded with the graph's zero row); the table is 8 wide while at most one row in 200 is
    longer than 8, else 16; `ell_slots` = the largest row length the table has to hold."""
    hub = R.regular_graph(1000, 7, 5)
    for v in range(10, 43):
        hub.add_edge(3, v)
    hs = [built.from_networkx(hub), built.from_networkx(R.regular_graph(600, 7, 6))]
    b = built.BatchArrays(hs)
    assert (b.ell_width, b.ell_slots, b.max_degree) == (8, 8, 40)
    assert b.ovf_ptr.dtype == np.int32 and b.ovf_ptr.shape == (1601,) and int(b.ovf_ptr[-1]) == 4
    assert b.ovf_ids.shape == (32,) and b.ovf_vals is None and b.ovf_max_blocks == 4
    for g, h in enumerate(hs):
        r0 = int(b.goff[g])
        for l in range(h.n):
            nb = h.col[h.rowptr[l]:h.rowptr[l + 1]].tolist()
            slots = [x for x in b.ell[r0 + l].tolist() if x < h.n]
            extra = b.ovf_ids[8 * b.ovf_ptr[r0 + l]:8 * b.ovf_ptr[r0 + l + 1]].tolist()
            assert sorted(slots) == sorted(nb[:8])
            assert [x for x in extra if x < h.n] == nb[8:] and all(x == h.n for x in extra[len(nb) - 8:] if len(nb) > 8)
    # weighted: overflow weights follow their ids, padding weighs 0
    for u, v in hub.edges():
        hub[u][v]["weight"] = 1 + (u + v) % 3
    hw = built.from_networkx(hub)
    bw = built.BatchArrays([hw])
    l = 3
    w_csr = hw.weight[hw.rowptr[l]:hw.rowptr[l + 1]]
    assert np.array_equal(bw.ovf_vals[:32], w_csr[8:].astype(np.float32))
    # widths
    cw = built.BatchArrays.choose_width
    assert cw(np.full(1000, 7)) == 8 and cw(np.full(1000, 8)) == 8 and cw(np.full(1000, 9)) == 16
    assert cw(np.r_[np.full(995, 7), np.full(5, 30)]) == 8          # 1 row in 200
    assert cw(np.r_[np.full(994, 7), np.full(6, 30)]) == 16
    assert cw(np.full(64, 20)) == 0                                  # overflow lists > 1/8 of the edges: row kernels
    d12 = built.BatchArrays([built.from_networkx(R.regular_graph(100, 12, 1))])
    assert (d12.ell_width, d12.ell_slots, d12.ovf_ptr) == (16, 12, None)
    d9 = built.BatchArrays([built.from_networkx(R.regular_graph(100, 9, 1)), built.from_networkx(R.regular_graph(50, 6, 1))])
    assert (d9.ell_width, d9.ell_slots) == (16, 9)
    # slots beyond ell_slots are padding in EVERY row (the kernels do not read them)
    assert (d12.ell[:, 12:] >= 100).all() and (d9.ell[:, 9:] >= np.repeat([100, 50], [100, 50])[:, None]).all()


def test_early_stopping_best_tracking_and_checkpoints(T, tmp_path, monkeypatch):
    """train_model's bookkeeping (TrainingNeural.py:421-482) with a scripted loss sequence."""
    monkeypatch.chdir(tmp_path)
    losses = iter([-10.0, -9.0, -8.0, -7.0, -20.0, -30.0])
    monkeypatch.setattr(T, "train_single_epoch", lambda *a, **k: next(losses))
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=8, number_epochs=6, patience=3, save_directory="m.pth", save_frequency=2)
    net, best, epoch, emb, hist = T.train_model({}, cfg)
    ref_stop, ref_best, ref_hist = R.early_stop_trace([-10.0, -9.0, -8.0, -7.0, -20.0, -30.0], cfg.tolerance, cfg.patience)
    assert (epoch, best, hist) == (ref_stop, ref_best, ref_hist) == (3, -10.0, [-10.0, -9.0, -8.0, -7.0])
    files = sorted(os.listdir(tmp_path))
    assert files == ["epoch_0_loss_-10.0000_m.pth", "epoch_2_loss_-8.0000_m.pth", "final_m.pth"]   # Q9 names
    ck = torch.load("final_m.pth", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "loss_history", "inputs", "config"}
    assert ck["config"].hidden_dim == 8 and tuple(ck["inputs"].shape) == (1000, 1000)
    net2, inputs, cfg2 = T.load_neural_model("final_m.pth", cfg)
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu(), net2.state_dict()[k].cpu())
    T.save_neural_model(net2, torch.optim.Adam(net2.parameters()), torch.nn.Embedding(4, 4), 7, [1.0], cfg, "again.pth")
    assert torch.load("again.pth", weights_only=False)["epoch"] == 7
    # README-style call: save_directory=None through kwargs -> nothing is saved (Q10)
    losses = iter([-1.0])
    monkeypatch.setattr(T, "open_file", lambda f: {})
    T.train_from_pickle("whatever.pkl", "model_x", number_epochs=1, save_directory=None, hidden_dim=8)
    assert not os.path.exists("final_model_x.pth")


def test_loss_helpers_match_oracle_values_and_gradients(T):
    ds = R.synthetic_dataset([(40, 5, 5)])
    (g, a_pad, nx_g, _t), = ds.values()
    torch.manual_seed(0)
    p = torch.softmax(torch.randn(40, 3), dim=1).requires_grad_(True)
    mine = T.compute_loss(T.apply_max_to_one_hot(T.override_fixed_nodes(p)), a_pad, 0, 2.0, 1000)
    mine.backward()
    p2 = p.detach().clone().requires_grad_(True)
    ref = R.cut_loss(R.straight_through_one_hot(R.override_terminals(p2)), a_pad, 2.0)
    ref.backward()
    assert float(mine) == float(ref) and torch.allclose(p.grad, p2.grad, atol=1e-6)
    with pytest.raises(RuntimeError):
        T.calculate_HC_vectorized(torch.eye(3)[:3], torch.zeros(3, 500))   # pad is hard-coded to 1000 (Q2)
    with pytest.raises(ValueError):
        T.extend_matrix_torch(torch.zeros(5, 5), 3)
    assert T.get_gnn is T.get_gnn_legacy and T.LoadNeuralModel is T.load_neural_model
    assert len(T.generate_terminal_permutations({"a": 0, "b": 1, "c": 2})) == 6


def test_reference_import_roots_resolve_to_this_package(built):
    built.install_compat()
    import Training.TrainingNeural as A
    import python.Training.TrainingNeural as B
    import DataGenerator.graphExtender as G
    import commons as Cm
    from gcn_max_cut_amd.Training import TrainingNeural as T
    assert A is T and B is T and hasattr(G, "process_graphs_from_folder") and hasattr(Cm, "open_file")
    blob = pickle.dumps(A.TrainingConfig(hidden_dim=12))
    assert pickle.loads(blob).hidden_dim == 12


def test_sharding_helpers(built):
    from gcn_max_cut_amd.engine import flat_layout, shard_for_rank
    for n, w in [(160, 8), (20, 3), (5, 8), (0, 2)]:
        parts = [list(shard_for_rank(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n)) and max(map(len, parts)) - min(map(len, parts)) <= 1
    offs, total = flat_layout(1000, 500, 3)
    assert offs == [0, 500000, 500500, 502000, 502003] and total == 502003
    # mixed sizes: contiguous runs balanced by edges, smallest possible largest run; equal weights == the count split
    from gcn_max_cut_amd.engine import shard_by_weight
    for n, w in [(160, 8), (20, 3), (5, 8), (0, 2), (1, 2)]:
        assert [shard_by_weight([7000] * n, r, w) for r in range(w)] == [shard_for_rank(n, r, w) for r in range(w)]
    split = lambda ws, w: [list(shard_by_weight(ws, r, w)) for r in range(w)]
    assert split([420, 80, 88, 96], 2) == [[0], [1, 2, 3]]            # by count it would be [0, 1] | [2, 3]
    assert split([10, 10, 10, 100, 10, 10], 3) == [[0, 1, 2], [3], [4, 5]]
    assert split([5, 1], 4) == [[0], [1], [], []]
    rng = np.random.RandomState(0)
    for _ in range(50):
        ws = rng.randint(1, 1000, size=rng.randint(1, 40)).tolist()
        w = int(rng.randint(1, 9))
        parts = split(ws, w)
        assert sum(parts, []) == list(range(len(ws)))
        best = max(sum(ws[i] for i in p) for p in parts)
        # optimality against brute force over contiguous partitions (dynamic programme)
        pre = np.r_[0, np.cumsum(ws)]
        dp = {(0, 0): 0}
        for k in range(1, w + 1):
            for j in range(len(ws) + 1):
                dp[(k, j)] = min(max(dp[(k - 1, i)], pre[j] - pre[i]) for i in range(j + 1) if (k - 1, i) in dp)
        assert best == dp[(w, len(ws))]


def test_bench_launcher_refuses_cleanly_without_gpus():
    """`bench.py --gpus N` (N > 1) spawns its own ranks; with fewer visible devices than ranks it must say so
    and exit non-zero BEFORE starting anything (here: no GPU at all)."""
    import subprocess, sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the launcher would really launch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "GCN_MAXCUT_BENCH_REHEARSE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "refusing to launch" in out.stderr and out.stdout.strip() == ""


def test_bench_helpers_fingerprint_and_graph_generation(monkeypatch, tmp_path):
    """`bench.csrc_fingerprint` names the kernel sources' CODE (the PMC records under profiles/ are quoted only while
    it matches): comments and blank lines do not move it, a code edit does.  `bench.generate` stays in-process under a
    profiler (forked workers hang once its library has initialised the GPU) and returns the same graphs either way."""
    import importlib, shutil, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    fp = bench.csrc_fingerprint()
    copy_root = tmp_path / "repo"
    shutil.copytree(os.path.join(root, "gcn-max-cut_amd", "csrc"), copy_root / "gcn-max-cut_amd" / "csrc",
                    ignore=shutil.ignore_patterns("build*", "*.o"))
    monkeypatch.setattr(bench, "ROOT", str(copy_root))
    assert bench.csrc_fingerprint() == fp
    f = copy_root / "gcn-max-cut_amd" / "csrc" / "head.hip"
    text = f.read_text()
    f.write_text("// a new comment\n\n/* and a block\n   comment */\n" + text.replace("// Per-graph", "//  Per-graph (edited)", 1))
    assert bench.csrc_fingerprint() == fp
    f.write_text(text + "\nstatic int a_line_of_code;\n")
    assert bench.csrc_fingerprint() != fp
    specs = [("reg", 40, 5, 7), ("gnp", 40, 0.2, 8), ("reg", 30, 4, 9)]
    monkeypatch.setenv("GCN_MAXCUT_BENCH_SERIAL_GEN", "1")
    serial = bench.generate(specs)
    monkeypatch.delenv("GCN_MAXCUT_BENCH_SERIAL_GEN")
    forked = bench.generate(specs)
    assert [sorted(g.edges()) for g in serial] == [sorted(g.edges()) for g in forked]


def test_sampler_semantics_are_numpy_version_independent(built, monkeypatch):
    """assign_partitions (TestingNeuralNetwork.py:18-46): host form and oracle use the pinned NumPy 1.x
    arithmetic (float64 running sum and compare) whatever NumPy is installed; on draws one ulp either side of
    the boundaries that differs from the NumPy >= 2 reading (float32 sum, float32 compare)."""
    from gcn_max_cut_amd.Testing import TestingNeuralNetwork as TN
    rng = np.random.RandomState(1)
    n = 44
    probs, draws = np.empty((n, 3), np.float32), []
    for l in range(n):
        a, b = rng.uniform(0.05, 0.45, 2)
        p0, p1 = np.float32(a), np.float32(b)
        probs[l] = [p0, p1, np.float32(1.0) - p0 - p1]
        if l >= 3:
            c0, c1_64, c1_32 = float(p0), float(p0) + float(p1), float(np.float32(p0 + p1))
            draws.append([np.nextafter(c0, 0.0), c0, np.nextafter(c1_64, 0.0), c1_64, 0.5 * (c1_64 + c1_32),
                          np.nextafter(c0, 1.0)][(l - 3) % 6])
    it = iter(draws)
    monkeypatch.setattr(np.random, "rand", lambda: next(it))
    host = TN.assign_partitions(probs)
    it2, it3 = iter(draws), iter(draws)
    ref = R.sample_partition(probs, lambda: next(it2))
    np2 = R.sample_partition_numpy2(probs, lambda: next(it3))
    assert host == ref and ref != np2
    # kind 0 (draw just below the first boundary): double compare -> class 0; float32 compare -> not class 0
    assert ref[3] == 0 and np2[3] != 0
