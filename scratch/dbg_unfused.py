import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, contextlib, io
import gcn_max_cut_amd as pkg
from gcn_max_cut_amd.Training import TrainingNeural as T
from tests import util
from oracle import c_oracle as CO
lib = pkg.hip.load()
for hidden, specs in ((16, [(100, 7, 1000), (50, 6, 1001), (64, 8, 1002), (30, 5, 1003)]), (500, [(1000, 7, 5), (1000, 7, 6)]), (500, [(500, 7, 7), (300, 6, 8)]), (64, [(120, 12, 11)])):
    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=hidden); torch.manual_seed(0)
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    params = util.np_params(net.state_dict())
    with contextlib.redirect_stdout(io.StringIO()):
        ds = util.product_dataset(specs)
    eng = net.engine(); items = list(ds.values())
    batch = pkg.GraphBatch([it[0] for it in items], None, eng.device)
    ref = np.concatenate([CO.forward(rp, cl, vl, params["conv1.weight"], params["conv1.bias"], params["conv2.weight"], params["conv2.bias"])["P"] for rp, cl, vl in util.csrs_of(ds)])
    out = {}
    for fuse in (1, 0):
        lib.gmc_set_fuse(fuse)
        P, S, loss = eng.forward(batch, 1.0, want_loss=True); torch.cuda.synchronize()
        out[fuse] = P.cpu().numpy()
        print("hidden", hidden, "fuse", fuse, "max |P - oracle|", np.abs(out[fuse] - ref).max())
    lib.gmc_set_fuse(1)
