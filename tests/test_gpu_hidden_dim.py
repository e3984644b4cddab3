"""Any hidden_dim >= 1 (TrainingNeural.py:42,66-67: TrainingConfig takes any int; n_nodes = 50 derives 25): the kernels
work on 16-byte column groups, the engine pads the hidden dimension to a multiple of 4 inside its flat buffer and the
module / state_dict keep the reference's shapes."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(built):
    built.hip.require_gpu()
    return built


def model(hidden, n_nodes=1000):
    from gcn_max_cut_amd.Training import TrainingNeural as T
    cfg = T.TrainingConfig(n_nodes=n_nodes, hidden_dim=hidden)
    torch.manual_seed(0)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    return T, cfg, net, embed, opt, util.np_params(net.state_dict())


@pytest.mark.parametrize("hidden", [250, 25, 7, 1, 498])
def test_step_at_hidden_dims_that_are_not_multiples_of_four(pkg, hidden):
    T, cfg, net, embed, opt, params = model(hidden)
    assert params["conv1.weight"].shape == (1000, hidden) and params["conv2.weight"].shape == (hidden, 3)
    specs = [(1000, 7, 401), (300, 8, 402), (120, 12, 403)] if hidden >= 25 else [(200, 7, 404), (90, 6, 405)]
    ds = util.product_dataset(specs)
    eng, tags = util.check_step_against_oracle(pkg, net, ds, params)
    assert eng.F == hidden and eng.Fp % 4 == 0 and 0 <= eng.Fp - hidden < 4
    sd = net.state_dict()
    assert tuple(sd["conv1.weight"].shape) == (1000, hidden) and tuple(sd["conv1.bias"].shape) == (hidden,)
    assert tuple(sd["conv2.weight"].shape) == (hidden, 3)
    for k, v in eng.views().items():          # the module's parameters ARE the engine's (logical) views
        assert sd[k].data_ptr() == v.data_ptr() and sd[k].stride() == v.stride()


def test_default_config_of_a_50_node_problem_trains_and_pads_stay_zero(pkg, tmp_path, monkeypatch):
    """TrainingConfig(n_nodes=50) -> dim_embedding 50, hidden_dim 25 (TrainingNeural.py:62-67): the reference
    schedule and the batched step both follow the C oracle; the pad column of W1 / b1 and pad row of W2 stay 0."""
    monkeypatch.chdir(tmp_path)
    from gcn_max_cut_amd.Training import TrainingNeural as T
    cfg = T.TrainingConfig(n_nodes=50)
    assert (cfg.dim_embedding, cfg.hidden_dim) == (50, 25)
    torch.manual_seed(0)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    params = util.np_params(net.state_dict())
    specs = [(50, 7, 411), (40, 6, 412), (48, 5, 413)]
    ds = util.product_dataset(specs, max_nodes=50)
    ct = CO.CTrainer(params, lr=cfg.learning_rate)
    csrs = util.csrs_of(ds)
    for epoch in range(3):
        ref = sum(float(ct.step([c])[0]) for c in csrs)
        got = T.train_single_epoch(ds, net, opt, embed, cfg)
        assert abs(got - ref) <= 1e-3 * max(1.0, abs(ref)), (epoch, got, ref)
    for epoch in range(2):
        ref = float(ct.step(csrs).sum())
        got = T.train_single_epoch(ds, net, opt, embed, cfg, graphs_per_step=3)
        assert abs(got - ref) <= 1e-3 * max(1.0, abs(ref)), (epoch, got, ref)
    eng = net.engine()
    pv = eng.padded_views()
    assert eng.Fp == 28
    assert float(pv["conv1.weight"][:, 25:].abs().max()) == 0.0 and float(pv["conv1.bias"][25:].abs().max()) == 0.0
    assert float(pv["conv2.weight"][25:].abs().max()) == 0.0
    after = ct.unpack()
    for k, v in net.state_dict().items():   # (single entries may differ by a step where a gradient's sign is fp32 noise)
        assert float(np.median(np.abs(v.cpu().numpy() - after[k]))) < 1e-5, k
    # checkpoint round trip keeps the logical shapes
    T.save_neural_model(net, opt, embed, 1, [0.0], cfg, "m.pth")
    net2, _inputs, _cfg = T.load_neural_model("m.pth", cfg)
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu(), net2.state_dict()[k].cpu()) and v.shape == net2.state_dict()[k].shape
    res = T.evaluate_model(net2, ds, cfg)
    assert res["num_samples"] == 3
