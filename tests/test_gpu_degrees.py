"""GPU parity tests of the degree handling (round 3): 16-slot neighbour tables at n = 1000, live-slot
specialisations (d = 9, 12, 14, 16), overflow lists for hub rows, G(n,p) graphs - all on the LDS-tiled FUSED kernels
(asserted from the kernel tags the library's probe reports), against the C oracle through the C ABI.
Reference: DataGenerator/GraphCreator.py:70-78,151-152 (regular graphs of degree min..max and G(n,p) graphs),
Training/TrainingNeural.py:363 (the default dataset is ..._d8_12_...)."""
import networkx as nx
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_dense as R
from tests import util

pytestmark = pytest.mark.gpu

FUSED = {"fwd1_fused", "head", "bwd1_fused", "finish"}


@pytest.fixture(scope="module")
def pkg(built):
    built.hip.require_gpu()
    return built


def model(pkg, hidden, seed=0):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=hidden)
    torch.manual_seed(seed)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    return T, cfg, net, embed, opt, util.np_params(net.state_dict())


def with_hub(n, d, seed, hub_degree, hub=5):
    g = R.regular_graph(n, d, seed)
    rng = np.random.RandomState(seed)
    others = [v for v in rng.permutation(n) if v != hub and not g.has_edge(hub, int(v))]
    for v in others[:hub_degree - d]:
        g.add_edge(hub, int(v), weight=1, capacity=1)
    assert g.degree(hub) == hub_degree
    return g


def gnp(n, p, seed):
    """G(n,p) as GraphCreator.generate_graph('prob') builds it, re-drawn until no node is isolated (DGL's GraphConv
    refuses zero in-degree nodes)."""
    while True:
        tmp = nx.convert_node_labels_to_integers(nx.fast_gnp_random_graph(n, p, seed=seed))
        if min(dict(tmp.degree()).values()) > 0:
            break
        seed += 1000
    g = nx.Graph()
    g.add_nodes_from(sorted(tmp.nodes()))
    g.add_edges_from(tmp.edges)
    nx.set_edge_attributes(g, 1, "weight")
    nx.set_edge_attributes(g, 1, "capacity")
    return g


def terms_for(graphs):
    return {i: R.seeded_terminals(g.number_of_nodes(), 100 + i) for i, g in graphs.items()}


@pytest.mark.parametrize("d,slots", [(9, 9), (12, 12), (14, 14), (16, 16)])
def test_degrees_9_to_16_at_n1000_stay_on_the_lds_path(pkg, d, slots):
    """BASELINE-size graphs (n = 1000, hidden 500) of degree 9..16: the 16-slot table fits beside two 64 KB tiles
    (round 2: 176,896 B > 160 KiB sent such a batch to the row kernels, ~4x slower), the kernels read only the
    live slots, and forward / loss / gradient meet the same bar as d = 7."""
    T, cfg, net, embed, opt, params = model(pkg, 500)
    graphs = {0: R.regular_graph(1000, d, 70 + d), 1: R.regular_graph(1000, d, 170 + d), 2: R.regular_graph(600, d, 270 + d)}
    ds = util.dataset_of(graphs, terms_for(graphs))
    items = list(ds.values())
    host = pkg.graph.BatchArrays([it[0] for it in items])
    assert (host.ell_width, host.ell_slots, host.ovf_ptr) == (16, slots, None)
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags


def test_one_hub_row_costs_its_own_edges_not_the_batch(pkg):
    """A d = 7 batch at n = 1000 with ONE node of degree 40: the table stays 8 slots wide, the hub's neighbours beyond
    the eighth go to the overflow lists and the batch stays on the fused LDS kernels."""
    T, cfg, net, embed, opt, params = model(pkg, 500)
    graphs = {0: R.regular_graph(1000, 7, 801), 1: with_hub(1000, 7, 802, 40), 2: R.regular_graph(1000, 7, 803),
              3: with_hub(700, 7, 804, 23, hub=2)}        # (hub == a terminal-to-be: relabelled by the extender)
    ds = util.dataset_of(graphs, terms_for(graphs))
    items = list(ds.values())
    host = pkg.graph.BatchArrays([it[0] for it in items])
    assert host.ell_width == 8 and host.ell_slots == 8 and host.max_degree == 40
    assert host.ovf_ptr is not None and int(host.ovf_ptr[-1]) == 4 + 2          # ceil(32/8) + ceil(15/8) blocks
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags
    # forward-only entry point (evaluate_model's path) on the same batch
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    P, S, loss = eng.forward(batch, 1.0, want_loss=True)
    Wl = [params[k] for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias")]
    off = 0
    for (rp, cl, vl) in util.csrs_of(ds):
        n = len(rp) - 1
        assert np.abs(P[off:off + n].cpu().numpy() - CO.forward(rp, cl, vl, *Wl)["P"]).max() < 1e-4
        off += n


def test_long_overflow_lists_and_neighbouring_hub_rows(pkg):
    """The whole-wave gather of a hub row: lists of several rounds (a degree-100 node: 12 overflow blocks), several hub
    rows among the sixteen rows one wave owns in a pass (nodes 16..19), the last row of a graph (whose clamped
    duplicates in the passes beyond n must not add twice), and a hub in a graph smaller than the batch's largest."""
    T, cfg, net, embed, opt, params = model(pkg, 500)
    g1 = with_hub(1000, 7, 811, 100)
    for i, hubnode in enumerate((16, 17, 18, 19)):
        rng = np.random.RandomState(900 + i)
        for v in rng.permutation(1000):
            if g1.degree(hubnode) >= 20 + 9 * i:
                break
            if int(v) != hubnode and not g1.has_edge(hubnode, int(v)):
                g1.add_edge(hubnode, int(v), weight=1, capacity=1)
    g2 = with_hub(776, 7, 812, 30, hub=775)
    graphs = {0: R.regular_graph(1000, 7, 813), 1: g1, 2: g2}
    ds = util.dataset_of(graphs, terms_for(graphs))
    host = pkg.graph.BatchArrays([it[0] for it in ds.values()])
    assert host.ell_width == 8 and host.max_degree >= 100 and int(host.ovf_ptr[-1]) >= 12 + 4 * 2 + 3
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags


def test_more_overflow_blocks_than_threads_in_one_graph(pkg, monkeypatch):
    """A caller of the C ABI may hand over any table / overflow split.  Force a 16-slot table onto dense G(n,p) graphs
    (the host layer would pick the row kernels): ~45 neighbours per row -> 4 blocks per row, > 1,024 blocks per graph -
    the commit's copy loop beyond a thread's first block - on the fused kernels (small graphs leave the LDS for it)."""
    T, cfg, net, embed, opt, params = model(pkg, 64)
    monkeypatch.setattr(pkg.graph.BatchArrays, "choose_width", staticmethod(lambda degi: 16))
    graphs = {0: gnp(300, 0.15, 21), 1: gnp(280, 0.12, 22)}
    ds = util.dataset_of(graphs, terms_for(graphs))
    host = pkg.graph.BatchArrays([it[0] for it in ds.values()])
    assert host.ell_width == 16 and 1024 < host.ovf_max_blocks <= 4095, host.ovf_max_blocks
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags


def test_overflow_pointer_without_blocks_is_a_batch_without_lists(pkg, monkeypatch):
    """C-ABI callers may hand over `ovf_ptr` (all zeros) with `ovf_max_blocks == 0` and no `ovf_ids`: the library treats
    it as a batch without lists (plain fused kernels; the OVF flavours would read block 0 of the batch)."""
    T, cfg, net, embed, opt, params = model(pkg, 64)
    orig = pkg.graph.BatchArrays.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        assert self.ovf_ptr is None
        self.ovf_ptr = np.zeros(self.R + 1, np.int32)
        self.ovf_ids = self.ovf_vals = None
        self.ovf_max_blocks = 0

    monkeypatch.setattr(pkg.graph.BatchArrays, "__init__", init)
    graphs = {0: R.regular_graph(300, 7, 31), 1: R.regular_graph(200, 8, 32)}
    ds = util.dataset_of(graphs, terms_for(graphs))
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags


def test_gnp_graphs_with_overflow_rows_on_the_16_slot_path(pkg):
    """G(n,p) graphs (GraphCreator 'prob'): Poisson-like degrees, a few rows beyond 16 -> 16-slot table + overflow."""
    T, cfg, net, embed, opt, params = model(pkg, 500)
    graphs = {0: gnp(1000, 0.0095, 5), 1: gnp(1000, 0.010, 6), 2: gnp(400, 0.022, 7)}
    ds = util.dataset_of(graphs, terms_for(graphs))
    items = list(ds.values())
    host = pkg.graph.BatchArrays([it[0] for it in items])
    assert host.ell_width == 16 and host.max_degree > 16 and host.ovf_ptr is not None
    assert 0 < host.ovf_max_blocks <= 51          # what fits beside the n = 1000 tiles and a 16-slot table
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags
    # a graph with more overflow blocks than the spare LDS holds sends the batch to the row kernels - same results
    dense = {0: gnp(1000, 0.013, 8), 1: gnp(300, 0.04, 9)}
    ds2 = util.dataset_of(dense, terms_for(dense))
    host2 = pkg.graph.BatchArrays([it[0] for it in ds2.values()])
    assert host2.ell_width == 16 and host2.ovf_max_blocks > 51
    eng, tags = util.check_step_against_oracle(pkg, net, ds2, params)
    assert "fwd1_fused" not in tags and "agg_fwd" in tags


@pytest.mark.parametrize("hidden", [64, 500])
def test_weighted_edges_with_overflow_and_wide_tables(pkg, hidden):
    """Integer edge weights on graphs with hub rows, loss scale C = 2: weights AND overflow lists is the one combination
    the fused LDS kernels leave to the row kernels (an overflow block's weights would be a global load inside a gather);
    degree-12 graphs with weights and no hub stay fused (16-slot table, `vals` variant)."""
    T, cfg, net, embed, opt, params = model(pkg, hidden)
    rng = np.random.RandomState(3)
    for build, fused in ((lambda: {0: with_hub(300, 7, 11, 30), 1: R.regular_graph(200, 6, 12)}, False),
                         (lambda: {0: with_hub(300, 12, 13, 41), 1: R.regular_graph(260, 11, 14), 2: R.regular_graph(100, 9, 15)}, False),
                         (lambda: {0: R.regular_graph(300, 12, 16), 1: R.regular_graph(260, 11, 17)}, True)):
        graphs = build()
        for g in graphs.values():
            for u, v in g.edges():
                g[u][v]["weight"] = int(rng.randint(1, 4))
        ds = util.dataset_of(graphs, terms_for(graphs))
        eng, tags = util.check_step_against_oracle(pkg, net, ds, params, C=2.0)
        assert (set(tags) == FUSED) == fused, tags


def test_overflow_batches_take_the_row_kernels_when_the_fused_kernels_are_off(pkg):
    """gmc_set_fuse(0) (and dropout): the one-kernel-per-operation LDS kernels do not walk overflow lists, such a
    batch then runs on the row kernels - same results to rounding."""
    T, cfg, net, embed, opt, params = model(pkg, 128)
    graphs = {0: with_hub(500, 7, 21, 33), 1: R.regular_graph(400, 7, 22)}
    ds = util.dataset_of(graphs, terms_for(graphs))
    lib = pkg.hip.load()
    prev = lib.gmc_set_fuse(0)
    try:
        eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    finally:
        lib.gmc_set_fuse(prev)
    assert "fwd1_fused" not in tags and "agg_fwd" in tags and "dw1" in tags


def test_reference_schedule_on_mixed_degrees(pkg):
    """One Adam step per graph (TrainingNeural.py:371-386) over graphs whose batches of ONE take different kernel
    flavours (7 slots, 16 slots with 12 live, overflow): losses of two epochs follow the C oracle trainer."""
    T, cfg, net, embed, opt, params = model(pkg, 64)
    graphs = {0: R.regular_graph(200, 7, 31), 1: R.regular_graph(240, 12, 32), 2: with_hub(220, 7, 33, 29), 3: gnp(250, 0.05, 34)}
    ds = util.dataset_of(graphs, terms_for(graphs))
    ct = CO.CTrainer(params, lr=cfg.learning_rate)
    csrs = util.csrs_of(ds)
    for epoch in range(2):
        ref = sum(float(ct.step([c])[0]) for c in csrs)
        got = T.train_single_epoch(ds, net, opt, embed, cfg)
        assert abs(got - ref) <= 1e-3 * max(1.0, abs(ref)), (epoch, got, ref)


@pytest.mark.parametrize("hidden,build", [
    (128, lambda: {0: R.regular_graph(530, 12, 41), 1: R.regular_graph(520, 11, 42)}),          # 16 slots, FS = 32, 8 rows per thread
    (256, lambda: {0: R.regular_graph(270, 12, 43), 1: R.regular_graph(262, 9, 44)}),           # 16 slots, FS = 64, 8 rows per thread
    (128, lambda: {0: with_hub(530, 7, 45, 33), 1: R.regular_graph(520, 6, 46)}),               # 8 slots + overflow, FS = 32, 8 rows
    (256, lambda: {0: with_hub(270, 7, 47, 29), 1: R.regular_graph(262, 8, 48)}),               # 8 slots + overflow, FS = 64, 8 rows
    (128, lambda: {0: with_hub(530, 12, 49, 45), 1: R.regular_graph(520, 11, 50)}),             # 16 slots + overflow, FS = 32, 8 rows
    (64, lambda: {0: gnp(270, 0.045, 51), 1: gnp(200, 0.05, 52)}),                              # 16 slots + overflow, FS = 64, 8 rows
])
def test_eight_rows_per_thread_flavours_of_the_new_paths(pkg, hidden, build):
    """Graphs just below the LDS capacity of a slice width run 8 rows per thread (the ACC = 8 instantiations): the
    16-slot, live-slot and overflow flavours of round 3 at those shapes, fused, against the oracle."""
    T, cfg, net, embed, opt, params = model(pkg, hidden)
    graphs = build()
    ds = util.dataset_of(graphs, terms_for(graphs))
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert set(tags) == FUSED, tags
