#!/usr/bin/env python3
"""Benchmark of the message-passing hot path (message + aggregate + update) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--mode fwd|train]

One process per GPU (N>1: launched by torch.distributed.run, RCCL backend).  A "step" is one pass
of the hot path over one resident batch: T message-passing rounds of
EdgeNetwork message -> AdjMsgAgg -> GRUUpdate through models.basic_model.BasicModel
plus, in the default `--mode train`, the backward pass and the ONE RCCL all-reduce of the flat gradient
bucket (the data-parallel step BASELINE.json's multi-GPU config describes); the forward-only rate of the
same batch is measured in the same run and reported under "forward".  Inputs are resident
in HBM before the timed region.  Molecules shard by graph: every rank holds its own 100k-molecule
batch (weak scaling), no data-path collective.

Prints ONE JSON line (rank 0) with the whole-job edges/s, the aggregator's live-measured roofline
figures and, at N=1, the CPU baseline (the oracle's dense restatement of the reference path) timed
on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (mols per GPU, hidden, MP steps, size distribution, description)
    "c2": (100_000, 64, 3, "drug", "100k synthetic mols/GPU, ~30 atoms/60 edges, hidden=64, 3 MP steps, sum aggregator"),
    "c3": (100_000, 128, 5, "drug", "att_model: AttEdgeNetwork feature gate + AdjMsgAgg + GRU + MaskBatchNorm, hidden=128, 5 MP steps"),
    "c3a": (100_000, 128, 5, "drug", "att_model with AttMsgAgg (scalar attention over ALL pairs of the padded row, incl. non-bonded) + AttEdgeNetwork + GRU + MaskBatchNorm, hidden=128, 5 MP steps"),
    "c4": (125_000, 128, 3, "drug", "125k synthetic mols/GPU (1M over 8), hidden=128, 3 MP steps"),
    "c2h128": (100_000, 128, 3, "drug", "c2 graphs at hidden=128"),
    "c5": (50_000, 256, 3, "skewed", "50k mols 10-200 atoms, preferential attachment, hidden=256"),
    "c1": (1_024, 22, 3, "lipo", "Lipophilicity-shaped (configs[0]): ~1k mols of 10-50 atoms, 22 atom features, 3 MP steps; "
                                 "one batch of the whole set per step, and the reference's batches of 16 beside it"),
    "tiny": (2_000, 64, 3, "drug", "2k mols (plumbing check)"),
}
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TF = 157.3    # dense fp32 MFMA peak (MI355X_MICROARCH.md); the 1e-5 bar rules out plain bf16/xf32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE.json configs[1] (the headline); c3/c4/c5 = configs[2..4] shapes")
    ap.add_argument("--mode", default="train", choices=["fwd", "train"],
                    help="train (default) = forward + backward + gradient all-reduce; fwd = inference pass only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def pmc_traffic(workload):
    """Per-launch HBM bytes of the aggregator from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes; see
    tools/pmc_summary.py).  PMC collection needs the profiler, so it cannot happen inside this process; the
    figure is a property of (kernel, workload) and is reported with its source, or None when no pass exists."""
    path = os.path.join(REPO, "profiles", "r01_pmc_%s.json" % workload)
    try:
        with open(path) as f:
            d = json.load(f)
        for k, v in d["kernels"].items():
            if "segsum" in k and "bwd" not in k:
                return v["hbm_bytes"], "profiles/" + os.path.basename(path)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def usable_cores():
    """Host cores this process may really use: min(affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, int(os.environ.get("MPNN_CPU_THREADS", "16")))   # the GPU box's CPU share per GPU is 16


def cpu_baseline(mb, hidden, steps, mode, budget_s):
    """The oracle's dense padded CPU path (== the reference's op sequence, validated against the
    reference's own outputs in tests/test_oracle_golden.py), batches of 16 molecules (the
    reference's batch size, test_lipo.py:150), all host cores, until `budget_s` is spent."""
    from mpnn_amd import synth
    from mpnn_amd.models.basic_model import BasicModel      # parameters only: same init as the GPU model
    from oracle import dense_ref as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(317)
    model = BasicModel(hidden, 4, hidden, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                       message_steps=steps)
    params = {k: v.detach().clone().requires_grad_(mode == "train") for k, v in model.state_dict().items()}
    mfp, ufp = O.sub(params, "mf."), O.sub(params, "uf.")
    edges = 0
    nb = 0
    # the dense path holds a (B, N, N, mf, nf) tensor (plus its gradient and the pair messages when training):
    # keep one batch under ~16 GB by dropping the largest molecules of a batch (C5: 200 atoms at H = 256 is 10.5 GB
    # per molecule), and say so in `sample`
    budget_bytes = 16e9 / (3.0 if mode == "train" else 1.5)
    n_atoms = np.asarray(mb.n_atoms, dtype=np.int64)
    dropped = 0
    t0 = time.perf_counter()
    for b0 in range(0, mb.num_mols - 16, 16):
        ids = np.arange(b0, b0 + 16)
        ids = ids[np.argsort(n_atoms[ids], kind="stable")]
        while len(ids) and len(ids) * float(n_atoms[ids].max()) ** 2 * hidden * hidden * 4 > budget_bytes:
            ids = ids[:-1]
            dropped += 1
        if not len(ids):
            continue
        ids = np.sort(ids)
        d = synth.to_dense(synth.select(mb, ids))
        afm, bfm, adj, mask = (torch.from_numpy(d[k]) for k in ("afm", "bfm", "adj", "mask"))
        ctx = torch.enable_grad() if mode == "train" else torch.no_grad()
        with ctx:
            h = afm
            for i in range(steps):
                # per step: message (edge_embed reused after step 0, as reuse_graph_tensors does) -> sum -> GRU
                if i == 0:
                    A = O.edge_matrices(mfp, bfm, hidden, hidden)
                pair = torch.einsum("bijmn,bjn->bijm", A, afm)
                h = O.gru_update(ufp, O.agg_adj(pair, adj), h, mask)
            if mode == "train":
                h.sum().backward()
        edges += int(adj.sum().item()) * steps
        nb += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": edges / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": "%d batches of 16 molecules of the same synthetic set%s, dense padded torch-CPU path, "
                      "%s, %.1f s" % (nb, " (%d molecules too large for the dense path left out)" % dropped if dropped else "",
                                      "forward+backward" if mode == "train" else "forward", dt)}


def stream_calibration(dev, nbytes):
    """What a plain 2-reads-1-write stream of the aggregator's byte count reaches on THIS box (torch `add`), so the
    aggregator's fraction of the 8 TB/s spec number can be read against the practical ceiling."""
    n = int(nbytes // 12)
    a = torch.empty(n, device=dev).normal_()
    b = torch.empty(n, device=dev).normal_()
    c = torch.empty(n, device=dev)
    for _ in range(3):
        torch.add(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.add(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return {"op": "torch.add, 2 reads : 1 write, same bytes", "GB/s": 12.0 * n / (ms * 1e-3) / 1e9, "ms": ms}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MPNN_DIST_BACKEND", "nccl")      # "gloo": rehearsal with ranks sharing one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from mpnn_amd import ops, synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel

    mols, hidden, T, dist_name, desc = WORKLOADS[args.workload]
    mb = synth.make_molecules(mols, hidden, seed=317 + rank, dist=dist_name, edge_features=4)
    graph = MolGraph.from_molbatch(mb, dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    mask = torch.ones(mb.num_atoms, 1, device=dev)
    V, E = graph.num_nodes, graph.num_edges
    graph.order, graph.type_ptr                          # index arrays built once, outside the timed region
    torch.manual_seed(317)                               # same weights on every rank
    if args.workload in ("c3", "c3a"):
        from mpnn_amd.models.att_model import BasicModel as AttModel
        from mpnn_amd.mpnn_functions import AdjMsgAgg, AttMsgAgg, GraphLevelOutput
        # the metric times message + aggregate + update; the model's default Set2Vec readout (100 LSTM steps over
        # every atom) is outside it, so the cheap readout closes the loss here as in the other workloads
        model = AttModel(hidden, 4, hidden, 1 if args.workload == "c3a" else 50, 8, message_opts={}, agg_opts={},
                         update_opts={}, readout_opts={}, message_steps=T, readout_func=GraphLevelOutput,
                         message_agg_func=AttMsgAgg if args.workload == "c3a" else AdjMsgAgg).to(dev)
    else:
        model = BasicModel(hidden, 4, hidden, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                           message_steps=T).to(dev)
    from mpnn_amd import parallel
    hot = [p for n, p in model.named_parameters() if not n.startswith("of.")]   # readout is off the hot path
    graph.transpose, graph.edge_dst
    bucket = parallel.GradientBucket(hot)
    total_mols = parallel.global_count(mols, dev)

    def step_fwd():
        with torch.no_grad():
            state, _ = model.message_passing(afm, graph, graph, mask)
        return state

    # d(loss)/d(state) of loss = sum(state) / total_mols (this shard's share of a global mean loss), built once: what
    # a readout's backward would hand to the path; seeding with it skips a V x H reduce, expand and scale per step
    seed = torch.full((V, hidden), 1.0 / float(total_mols), device=dev) if args.mode == "train" else None

    def step_train():
        bucket.zero()
        state, _ = model.message_passing(afm, graph, graph, mask)
        state.backward(gradient=seed.view_as(state))
        bucket.all_reduce()                              # ONE RCCL all-reduce of the flat gradient bucket
        return state

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, timer):
        """W untimed steps, then exactly K steps between barrier+synchronize fences; max over ranks."""
        for _ in range(args.warmup):
            step()
        ops.set_kernel_timer(timer)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        ops.set_kernel_timer(None)
        t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item())

    edges = torch.tensor([float(E)], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(edges)
    total_edges = float(edges.item())
    timer = ops.KernelTimer(["segsum", "edge_message", "gru_update"])
    if args.mode == "train":
        dt_fwd = timed(step_fwd, None)
        dt = timed(step_train, timer)
    else:
        dt_fwd = None
        dt = timed(step_fwd, timer)

    batch16 = None
    if args.workload == "c1" and world == 1:
        # the reference driver's own batching (test_lipo.py:150): 64 batches of 16 molecules per epoch, each with
        # its own CSR; launch-latency-bound on a GPU, reported beside the whole-set batch
        parts = []
        for b0 in range(0, mols, 16):
            sub = synth.select(mb, np.arange(b0, min(b0 + 16, mols)))
            gsub = MolGraph.from_molbatch(sub, dev)
            gsub.order, gsub.type_ptr, gsub.transpose, gsub.edge_dst
            a = torch.from_numpy(sub.atom_feat).to(dev)
            parts.append((a, gsub, torch.ones(a.shape[0], 1, device=dev)))

        def epoch_fwd():
            with torch.no_grad():
                for a, gs, mk in parts:
                    model.message_passing(a, gs, gs, mk)

        def epoch_train():
            for a, gs, mk in parts:
                bucket.zero()
                state, _ = model.message_passing(a, gs, gs, mk)
                (state.sum() / 16).backward()

        batch16 = (timed(epoch_fwd, None), timed(epoch_train, None))

    hoisted = None
    if world == 1 and hasattr(model, "hoist_message"):
        model.hoist_message = True
        hoisted = (timed(step_fwd, None), timed(step_train, None) if args.mode == "train" else float("nan"))
        model.hoist_message = False

    if rank == 0:
        F = hidden
        seg_ms = timer.mean_ms("segsum")
        weighted = graph.agg_weight is not None
        alg_bytes = 4.0 * F * (E + V) + 4.0 * (V + 1) + (4.0 * E if weighted else 0.0)   # msg rows + out rows + row_ptr (+ weights)
        achieved = alg_bytes / (seg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload)
        out = {
            "metric": "edges/sec (message+aggregate+update)",
            "value": total_edges * T * args.steps / dt,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "mode": ("train: forward + backward + flat-gradient all-reduce" if args.mode == "train" else "forward only"), "mols_per_gpu": mols,
                       "atoms_per_gpu": V, "edges_per_gpu": E, "hidden": hidden, "mp_steps": T,
                       "edge_features": 4, "edge_types": graph.num_types, "parallelism": "dp%d" % world,
                       "edges_counted": "directed edges x MP steps per pass",
                       "math": ("fp32 matrix pipe (MPNN_GRU_MATH=fp32)" if os.environ.get("MPNN_GRU_MATH") == "fp32" else
                                "fp32 data and accumulation; dense contractions as three-way bf16 operand splits "
                                "(six bf16 MFMAs per fp32 product), parity 1e-5 as the fp32 kernels")},
            "roofline": {"kernel": "segsum_pair_kernel (aggregator, mpnn_segsum_f32)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": seg_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "kernels_ms": {k: timer.mean_ms(k) for k in ("edge_message", "segsum", "gru_update")},
        }
        # the two dense contractions of the path against the fp32 matrix-core peak (fp32 MFMA 157.3 TF).
        # They run as bf16x6 (three-way split operands, six bf16 MFMAs per fp32 product), so "achieved" is in
        # fp32-EQUIVALENT flops: useful multiply-adds of the fp32 problem, not bf16 issue slots.
        msg_ms, gru_ms = timer.mean_ms("edge_message"), timer.mean_ms("gru_update")
        out["roofline_contractions"] = [
            {"kernel": "typed edge message (mpnn_edge_message_f32)", "bound": "mfma", "unit": "TFLOP/s",
             "peak": MFMA_F32_PEAK_TF, "achieved": 2.0 * F * F * E / (msg_ms * 1e-3) / 1e12,
             "frac": 2.0 * F * F * E / (msg_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, "avg_launch_ms": msg_ms,
             "hbm_GBs_algorithmic": (4.0 * F * V + 4.0 * F * E + 8.0 * E) / (msg_ms * 1e-3) / 1e9},
            {"kernel": "masked GRU update (mpnn_gru_update_f32)", "bound": "mfma", "unit": "TFLOP/s",
             "peak": MFMA_F32_PEAK_TF, "achieved": 12.0 * F * F * V / (gru_ms * 1e-3) / 1e12,
             "frac": 12.0 * F * F * V / (gru_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, "avg_launch_ms": gru_ms},
        ]
        if world == 1:
            out["roofline"]["stream_calibration"] = stream_calibration(dev, alg_bytes)
        if dt_fwd is not None:
            out["forward"] = {"value": total_edges * T * args.steps / dt_fwd, "unit": "edges/s",
                              "ms_per_step": dt_fwd / args.steps * 1e3,
                              "note": "same batch, inference pass only (no backward, no all-reduce)"}
        if batch16 is not None:
            out["batches_of_16"] = {
                "train_edges_per_s": total_edges * T * args.steps / batch16[1],
                "forward_edges_per_s": total_edges * T * args.steps / batch16[0],
                "note": "same molecules stepped through as 64 batches of 16 (the reference driver's batch size); "
                        "one optimizer-sized step per batch, ~60 kernel launches each"}
        if hoisted is not None:
            out["hoisted_message"] = {
                "train_ms_per_step": hoisted[1] / args.steps * 1e3, "forward_ms_per_step": hoisted[0] / args.steps * 1e3,
                "note": "NOT the headline: BasicModel.hoist_message=True computes message+aggregate once per pass "
                        "instead of once per MP step (their input is the constant afm, models/basic_model.py:57, so the "
                        "result is bit-identical); reported to show what the reference's own structure leaves on the table"}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(mb, hidden, T, args.mode, args.cpu_seconds)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
