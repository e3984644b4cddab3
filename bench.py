#!/usr/bin/env python3
"""Headline benchmark: GCN max-cut training epochs/sec + SpMM achieved HBM GB/s on
n=1000 d=7 regular graphs (BASELINE.json), one process per GPU.

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (forward, cut loss, backward, RCCL gradient all-reduce
when N > 1, fused Adam) over this rank's batch of graphs; inputs are resident in HBM before
the timed region.  Workload at N=1: config[3] of BASELINE.json on one GPU - 160 synthetic
n=1000 d=7 graphs (R = 160,000 rows, 646 MB algorithmic bytes per F=500 SpMM, beyond the
256 MiB Infinity Cache).  `value` counts epochs of 160 graphs per second.  --scaling weak
(default): every rank holds its own 160 graphs (N*160 graphs per step); --scaling strong: the 160
graphs are split over the ranks (160/N each - BASELINE config[3] literally at N=8).  With N > 1 the
line of a weak run also carries `strong_scaling`, the same job re-timed on the strong split.

The JSON line also carries
  roofline      - the SpMM the metric names (LDS-tiled CSR SpMM of the layer-1 aggregation,
                  forward + backward launches): algorithmic bytes / mean launch duration, HIP
                  events recorded by the library on the launch stream during a
                  region of K steps run as one kernel per operation (gmc_set_fuse(0));
  roofline_step - the dominant kernel of the TIMED (default, fused) step, priced on the bytes it
                  must move (compulsory) with the PMC-measured traffic beside them; the other fused
                  kernel under `also`;
  sequential    - the reference's own schedule (one Adam step per graph, TrainingNeural.py:371-386)
                  on the same 160 graphs, one hipGraph per epoch: the like-for-like partner of
  cpu_baseline  - the CPU oracle timed on this box's host cores on a bounded sample, rank 0, N=1 only:
                  variant A = oracle/ref_dense.py (reference-structured torch-CPU port: dense [n,1000]
                  GEMM, per-row Python one-hot, dense loss, one Adam step per graph) at torch's default
                  thread count, at 16 and at 1 thread (`value` = the best of them, `cores` says which);
                  variant B = oracle/gcn_oracle.c (scalar C: CSR SpMM, fused
                  argmax / loss), 1 thread.

Order of the regions inside one run: `sequential`, the per-kernel probe (150 untimed + K eager steps with HIP
events around every kernel), the one-kernel-per-operation leg, and LAST the headline: W untimed warmup steps,
then exactly K timed steps between barrier + synchronize pairs.  A fresh process needs ~30 steps to reach the
clocks the chip sustains; with the headline last its timed steps are sustained-clock steps at any K / W.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

EPOCH_GRAPHS = 160  # BASELINE.json configs[3]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--graphs-per-gpu", type=int, default=None,
                    help="default: 160 (weak scaling) or 160 / N (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 160 graphs per GPU; strong: 160 graphs in total (BASELINE configs[3] at N=8)")
    ap.add_argument("--nodes", type=int, default=1000, help="nodes per graph")
    ap.add_argument("--degree", type=int, default=7)
    ap.add_argument("--hidden", type=int, default=500)
    ap.add_argument("--mode", choices=["batched", "sequential"], default="batched",
                    help="batched: one Adam step per rank-batch (DP schedule); sequential: the "
                         "reference's one Adam step per graph (N=1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of EACH CPU baseline sample (three are timed)")
    ap.add_argument("--no-sequential", action="store_true", help="skip the reference-schedule region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="do not record per-kernel HIP events")
    return ap.parse_args()


def regular_graph(n, d, seed):
    """Synthetic input of SURVEY section 8d: nx.random_regular_graph + unit weights, the
    reference's generator (DataGenerator/GraphCreator.py:70-90) restated."""
    import networkx as nx
    tmp = nx.convert_node_labels_to_integers(nx.random_regular_graph(d=d, n=n, seed=seed))
    g = nx.Graph()
    g.add_nodes_from(sorted(tmp.nodes()))
    g.add_edges_from(tmp.edges)
    nx.set_edge_attributes(g, 1, "weight")
    nx.set_edge_attributes(g, 1, "capacity")
    return g


def terminals_of(n, seed):
    import random
    return random.Random(seed).sample(range(n), 3)


def self_launch(args) -> int:
    """`bench.py --gpus N` without a rendezvous environment: start the N ranks as CHILD processes
    (torch.distributed.run) before this process has touched a GPU, hand their output through and return
    their exit status.  Nothing here initialises HIP (device_count() does not)."""
    import socket
    import subprocess
    rehearse = os.environ.get("GCN_MAXCUT_BENCH_REHEARSE") == "1"
    have = torch.cuda.device_count()
    if have < (1 if rehearse else args.gpus):
        print(f"bench.py: --gpus {args.gpus} but {have} HIP device(s) visible; refusing to launch "
              f"(set GCN_MAXCUT_BENCH_REHEARSE=1 to rehearse N ranks on one device over gloo)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal aid (one-GPU boxes): GCN_MAXCUT_BENCH_REHEARSE=1 puts every rank on device 0 and uses
    # gloo, to exercise the N > 1 control flow; never set by the driver
    rehearse = os.environ.get("GCN_MAXCUT_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    backend = None
    # GCN_MAXCUT_DP_SINGLE_RANK=1 (one-GPU boxes): ONE rank over RCCL runs the data-parallel step sequence -
    # communicator set-up, the collective on the launch stream and graph capture beside RCCL's threads
    single_dp = world == 1 and os.environ.get("GCN_MAXCUT_DP_SINGLE_RANK") == "1"
    dp = world > 1 or single_dp
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dp:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        backend = dist.get_backend()
        world = dist.get_world_size()      # what the collective library saw, not what the flag said
    if args.mode == "sequential" and world > 1:
        raise SystemExit("sequential mode is the single-GPU reference schedule")
    if args.graphs_per_gpu is None:
        args.graphs_per_gpu = EPOCH_GRAPHS if args.scaling == "weak" else max(1, EPOCH_GRAPHS // world)

    import gcn_max_cut_amd as pkg
    from gcn_max_cut_amd.DataGenerator import graphExtender as GE
    from gcn_max_cut_amd.Training import TrainingNeural as T

    gpg, n, d = args.graphs_per_gpu, args.nodes, args.degree
    seeds = [3000 + rank * gpg + i for i in range(gpg)]
    graphs = {i: regular_graph(n, d, s) for i, s in enumerate(seeds)}
    terms = {i: terminals_of(n, s) for i, s in enumerate(seeds)}
    with contextlib.redirect_stdout(io.StringIO()):
        dataset = GE.process_graphs_from_folder(graphs, terms, 1000)
    assert len(dataset) == gpg, "a synthetic graph was skipped by the terminal normalisation"

    cfg = T.TrainingConfig(n_nodes=1000, hidden_dim=args.hidden, number_epochs=10 ** 9, patience=10 ** 9)
    torch.manual_seed(0)  # CPU RNG: identical Xavier init on every rank
    net, embed, opt = T.setup_model_and_optimizer(cfg)
    gps = gpg if args.mode == "batched" else 1
    trainer = T.FusedTrainer(net, opt, cfg, graphs_per_step=gps, local_shard=True)
    net.train()
    trainer.prepare(dataset)
    eng = trainer.eng
    spmm_bytes = trainer._batches[0].spmm_bytes(args.hidden)

    def sync():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(tr, data, warmup, steps):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize pairs; max over ranks."""
        loss = None
        for _ in range(warmup):
            tr.epoch(data)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.epoch(data)
        sync()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if dp:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()), loss

    launches_per_step = 16 * len(trainer._batches)
    # Order of the regions: the reference-schedule run, the per-kernel probe and the one-kernel-per-operation leg
    # come FIRST, the headline's W warmup + K timed steps LAST: a freshly started process needs ~30 steps (7 ms)
    # to reach the clocks the chip sustains, and training throughput is a sustained quantity.
    # the reference's own schedule on the same graphs: one Adam step per graph (TrainingNeural.py:371-386),
    # one hipGraph per epoch - what `cpu_baseline` (same schedule, host cores) is to be compared with
    sequential = None
    if not dp and args.mode == "batched" and not args.no_sequential:
        torch.manual_seed(0)
        net_s, embed_s, opt_s = T.setup_model_and_optimizer(cfg)
        net_s.train()
        tr_q = T.FusedTrainer(net_s, opt_s, cfg, graphs_per_step=1, local_shard=True)
        k_seq = max(3, args.steps // 10)
        dt, _ = timed(tr_q, dataset, 2, k_seq)
        sequential = {"value": k_seq * gpg / EPOCH_GRAPHS / dt, "unit": f"epochs/s (1 epoch = {EPOCH_GRAPHS} graphs)",
                      "ms_per_epoch": 1e3 * dt / k_seq, "us_per_graph_step": 1e6 * dt / (k_seq * gpg),
                      "optimizer_steps_per_epoch": gpg, "epochs_timed": k_seq,
                      "schedule": "one Adam step per graph, dataset order (reference schedule), hipGraph per epoch"}

    # per-kernel means: the same K steps repeated with eager launches, every kernel bracketed by
    # HIP events on the launch stream (a graph replay cannot carry the probe's event records)
    kernels = {}
    eager_ms = None
    if not args.no_probe:
        trainer.allow_graph = False
        for _ in range(150):  # the probe's own untimed steps (kernel set-up; ~40 ms of full load bring the chip to its
                              # sustained clocks after the light reference-schedule region): its means feed `roofline_step`
            trainer.epoch(dataset)
        sync()
        te = time.perf_counter()
        with pkg.hip.Probe(launches_per_step * args.steps) as probe:
            for _ in range(args.steps):
                trainer.epoch(dataset)
            sync()
        eager_ms = 1e3 * (time.perf_counter() - te) / args.steps
        for tag, ms in probe.records:
            kernels.setdefault(tag, []).append(ms)
    kmean = {k: float(np.mean(v)) for k, v in kernels.items()}

    # third region (EVERY rank runs it: each step holds a collective when N > 1): the same step as
    # one kernel per operation -> timings of the stand-alone SpMM kernel
    spmm, k2 = [], {}
    if args.mode == "batched" and not args.no_probe:
        lib = pkg.hip.load()
        prev = lib.gmc_set_fuse(0)
        for _ in range(10):
            trainer.epoch(dataset)
        with pkg.hip.Probe(launches_per_step * args.steps) as p2:
            for _ in range(args.steps):
                trainer.epoch(dataset)
            sync()
        lib.gmc_set_fuse(prev)
        for tag, ms in p2.records:
            k2.setdefault(tag, []).append(ms)
        spmm = [ms for tag in ("agg_fwd", "agg_bwd") for ms in k2.get(tag, [])]

    trainer.allow_graph = True

    # timed region: the step's launches are replayed hipGraphs (one per step on one GPU; forward/backward and
    # Adam graphs around the eager all-reduce on N GPUs) - no per-launch host work
    elapsed, last_loss = timed(trainer, dataset, args.warmup, args.steps)

    # N > 1, weak run: the same job on the strong split (160 graphs in total, 160 / N per rank)
    strong = None
    if dp and args.scaling == "weak" and args.mode == "batched":
        per = min(len(dataset), max(1, EPOCH_GRAPHS // world))
        sub = {k: dataset[k] for k in list(dataset)[:per]}
        tr_s = T.FusedTrainer(net, opt, cfg, graphs_per_step=per, local_shard=True)
        dt, _ = timed(tr_s, sub, max(3, args.warmup // 2), args.steps)
        strong = {"graphs_per_gpu": per, "graphs_per_step": per * world, "ms_per_step": 1e3 * dt / args.steps,
                  "value": args.steps * per * world / EPOCH_GRAPHS / dt, "scaling": "strong"}

    # per-step collective time (events on the launch stream around the eager all-reduce), N > 1
    allreduce_ms = None
    if dp:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        sync()
        for a, b in evs:
            a.record()
            eng.allreduce_grad()
            b.record()
        sync()
        allreduce_ms = float(np.mean([a.elapsed_time(b) for a, b in evs[2:]]))

    if rank != 0:
        if dp:
            dist.destroy_process_group()
        return

    epochs = args.steps * world * gpg / EPOCH_GRAPHS
    out = {
        "metric": "training epochs/sec (n=1000 d=7 graphs)",
        "value": epochs / elapsed,
        "unit": f"epochs/s (1 epoch = {EPOCH_GRAPHS} graphs)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{gpg} x n={n} d={d} regular graphs per GPU (BASELINE configs[3]"
                        f"{' on one GPU' if world == 1 else ''}; {world * gpg} graphs per step), "
                        f"GCN 1000->{args.hidden}->3, {args.mode} steps, Adam lr 1e-3",
            "graphs_per_gpu": gpg, "rows_per_gpu": gpg * n, "nnz_per_gpu": gpg * n * d,
            "hidden_dim": args.hidden, "mode": args.mode, "epoch_graphs": EPOCH_GRAPHS,
            "optimizer_steps_per_step": len(trainer._batches),
            "parallelism": f"dp{world}" if dp else "single",
        },
        "backend": backend,
        "last_loss": last_loss,
        "launch": ("hipGraph replay" if trainer._graph is not None else
                   "hipGraphs around the eager all-reduce" if trainer._dp_graph is not None else
                   "eager launches queued behind the previous step's backward (losses stored into pinned host memory)"
                   if getattr(trainer, "_loss_host_dev", None) and not dp else
                   "eager launches around the all-reduce (loss published to pinned host memory ahead of Adam)"
                   if dp and getattr(trainer, "_step_host_dev", None) else "eager"),
        "ms_per_step_eager_probed": eager_ms,
        "kernels_ms": {k: round(v, 5) for k, v in sorted(kmean.items())},
        "steady_state_note": "the first ~30 steps of a fresh process run ~9 % slower (clock ramp); the timed steps are the "
                             "LAST region of this run (after the reference-schedule, probe and one-kernel-per-op regions), "
                             "so they are sustained-clock steps at any --steps/--warmup; DESIGN.md quotes the defaults "
                             "--steps 200 --warmup 30",
    }
    if allreduce_ms is not None:
        out["allreduce_ms_per_step"] = allreduce_ms
    if strong is not None:
        out["strong_scaling"] = strong
    if sequential is not None:
        out["sequential"] = sequential

    # dominant kernels of the TIMED step on the bytes they must move (compulsory), PMC traffic beside them
    R_, F_, N_ = gpg * n, args.hidden, 1000
    chunks = 8
    tfile_f = os.path.join(ROOT, "profiles", "traffic_fused.json")
    traffic_f = json.load(open(tfile_f)) if os.path.exists(tfile_f) else {}
    fused = {}
    for tag, compulsory, what in (
            ("fwd1_fused", R_ * F_ * 4 + N_ * F_ * 4 + R_ * 16 + R_ * 12 * 8,
             "H write + W1 + neighbour table + 8 Zpart partials"),
            ("bwd1_fused", R_ * F_ * 4 + R_ * 16 + R_ * 16 + chunks * N_ * F_ * 4,
             "H read + (GY2, dinv) rows + neighbour table + 8 dW1 chunk partials")):
        if tag in kmean:
            dur = kmean[tag] * 1e-3
            rec = traffic_f.get(tag) if traffic_f.get("rows") == R_ and traffic_f.get("F") == F_ else None
            fused[tag] = {"kernel": {"fwd1_fused": "fwd1_lds_kernel", "bwd1_fused": "bwd1_reg_kernel"}[tag], "bound": "hbm",
                          "achieved": compulsory / dur / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": compulsory / dur / 1e9 / HBM_PEAK_GBS,
                          "traffic": rec, "compulsory_bytes_per_launch": compulsory, "compulsory_bytes_are": what,
                          "mean_launch_us": dur * 1e6}
    if fused:
        dom = max(fused, key=lambda k: fused[k]["mean_launch_us"])
        out["roofline_step"] = dict(fused[dom], also={k: v for k, v in fused.items() if k != dom})
    else:
        out["roofline_step"] = None

    if k2:
        out["kernels_ms_unfused"] = {k: round(float(np.mean(v)), 5) for k, v in sorted(k2.items())}
    if spmm and args.mode == "batched":
        dur = float(np.mean(spmm)) * 1e-3
        achieved = spmm_bytes / dur / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_spmm.json")
        if os.path.exists(tfile):
            rec = json.load(open(tfile))
            if rec.get("rows") == gpg * n and rec.get("F") == args.hidden:
                traffic = rec.get("hbm_bytes_per_launch")
        out["roofline"] = {
            "kernel": "spmm_lds_kernel (layer-1 aggregation SpMM, fwd + bwd launches; one-kernel-per-op leg)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": spmm_bytes, "mean_launch_us": dur * 1e6,
            "launches": len(spmm),
        }
    else:
        out["roofline"] = None

    if not dp and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, graphs, terms, net)
        if sequential is not None:   # same schedule on both sides: GPU reference-schedule rate / CPU rate
            out["cpu_baseline"]["gpu_sequential_over_cpu"] = sequential["value"] / out["cpu_baseline"]["value"]
        out["parity"] = parity_gate(pkg, T, net, dataset)
    print(json.dumps(out), flush=True)
    if dp:
        dist.destroy_process_group()


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, graphs, terms, net):
    """The oracle as the CPU path on the first graphs of the same workload, reference schedule (one Adam
    step per graph, TrainingNeural.py:371-386).  Variant A: oracle/ref_dense.py, reference-structured
    torch-CPU (dense [n,1000] GEMM, per-row Python one-hot, dense loss, autograd, torch Adam) at all
    threads and at one thread.  Variant B: oracle/gcn_oracle.c, scalar C with CSR SpMM and fused
    argmax / loss (so the comparison is not inflated by the reference's dense waste), one thread."""
    from oracle import c_oracle as CO
    from oracle import ref_dense as R
    import copy
    all_threads = torch.get_num_threads()
    params = R.init_params(1000, args.hidden, 3, seed=0)

    def run_a(threads):
        torch.set_num_threads(threads)
        tr = R.Trainer(params, lr=1e-3)
        t_total, k = 0.0, 0
        for i in sorted(graphs):
            if t_total > args.cpu_seconds or k >= 64:
                break
            g = copy.deepcopy(graphs[i])  # bench graphs were already relabelled in place
            item = [R.graph_from_networkx(g), R.dense_adjacency(g, 1000), g, [0, 1, 2]]
            t0 = time.perf_counter()
            tr.step([item])
            t_total += time.perf_counter() - t0
            k += 1
        torch.set_num_threads(all_threads)
        return t_total, k

    def run_b():
        ct = CO.CTrainer({k: v.numpy() for k, v in params.items()}, lr=1e-3)
        t_total, k = 0.0, 0
        for i in sorted(graphs):
            if t_total > args.cpu_seconds or k >= 160:
                break
            csr = CO.csr_of(graphs[i])
            t0 = time.perf_counter()
            ct.step([csr])
            t_total += time.perf_counter() - t0
            k += 1
        return t_total, k

    def rec(t_total, k, cores, what):
        per = t_total / max(k, 1)
        return {"value": 1.0 / (per * EPOCH_GRAPHS), "cores": cores, "ms_per_graph_step": per * 1e3,
                "sample": f"{k} sequential graph-steps (n={args.nodes} d={args.degree}, hidden {args.hidden}) of {what}, {t_total:.1f} s"}

    what_a = "the reference-structured torch-CPU oracle (variant A)"
    a_all = rec(*run_a(all_threads), all_threads, what_a)
    mid = min(16, all_threads)   # a GPU box hands one GPU's job a share of the host: many more threads than that oversubscribe
    a_mid = rec(*run_a(mid), mid, what_a) if mid not in (1, all_threads) else None
    a_one = rec(*run_a(1), 1, what_a)
    b_one = rec(*run_b(), 1, "the scalar C oracle: CSR SpMM, fused argmax/loss (variant B)")
    variants = {"A_all_threads": a_all, "A_1_thread": a_one, "B_sparse_c_1_thread": b_one}
    if a_mid is not None:
        variants[f"A_{mid}_threads"] = a_mid
    best = max((v for k, v in variants.items() if k.startswith("A_")), key=lambda v: v["value"])
    return {
        # the reference-structured variant at the thread count that serves it best (all of them are listed)
        "value": best["value"], "unit": f"epochs/s (1 epoch = {EPOCH_GRAPHS} graphs)",
        "cores": best["cores"], "kind": "port", "sample": best["sample"],
        "schedule": "one Adam step per graph (reference schedule; compare with `sequential`, not with `value`)",
        "variants": variants,
        "host_cpus": os.cpu_count(), "torch_default_threads": all_threads, "cpu_model": cpu_model_name(),
        "torch": torch.__version__,
    }


def parity_gate(pkg, T, net, dataset):
    """Parity numbers reported with every perf number: probabilities
    vs the C oracle on two graphs of the workload with the current weights."""
    from oracle import c_oracle as CO
    params = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    worst, worst_unfused, arg_ok, loss_ok = 0.0, 0.0, True, True
    net.eval()
    for key in list(dataset)[:2]:
        g, a_pad, nx_g, _t = dataset[key]
        with torch.no_grad():
            P = net(g, a_pad).cpu().numpy()
        rp, cl, vl = CO.csr_of(nx_g)
        ref = CO.forward(rp, cl, vl, params["conv1.weight"], params["conv1.bias"], params["conv2.weight"],
                         params["conv2.bias"])["P"]
        S, loss, _ = CO.loss_grad(rp, cl, vl, ref)
        worst = max(worst, float(np.abs(P - ref).max()))
        # the one-kernel-per-operation sequence (whose SpMM the `roofline` object times) on the same graph
        lib = pkg.hip.load()
        prev = lib.gmc_set_fuse(0)
        try:
            with torch.no_grad():
                P0 = net(g, a_pad).cpu().numpy()
        finally:
            lib.gmc_set_fuse(prev)
        worst_unfused = max(worst_unfused, float(np.abs(P0 - ref).max()))
        arg_ok &= bool(np.array_equal(P.argmax(1)[3:], ref.argmax(1)[3:]))
        res = T.evaluate_model(net, {0: dataset[key]}, T.TrainingConfig())
        loss_ok &= res["total_loss"] == loss
    net.train()
    return {"max_abs_prob_diff": worst, "max_abs_prob_diff_one_kernel_per_op": worst_unfused, "argmax_equal": arg_ok,
            "loss_equals_minus_cut": loss_ok, "tolerance": 1e-4}


if __name__ == "__main__":
    main()
