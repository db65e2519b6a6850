#!/usr/bin/env python3
"""Benchmark of the message-passing hot path (message + aggregate + update) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--mode fwd|train]
                    [--scaling weak|strong]

One process per GPU.  With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this script starts its own N
ranks -- `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, before anything here
touches the GPU -- and relays the child's single JSON line; under the driver's launcher it is one of the ranks.

A "step" is one pass of the hot path over the rank's resident molecules: T message-passing rounds of
EdgeNetwork message -> AdjMsgAgg -> GRUUpdate through models.basic_model.BasicModel plus, in the default
`--mode train`, the backward pass and the ONE RCCL all-reduce of the flat gradient bucket (the data-parallel step
BASELINE.json's multi-GPU config describes); the forward-only rate of the same batch is measured in the same run and
reported under "forward".  Inputs are resident in HBM before the timed region.  Molecules shard by graph, no
data-path collective:

  --scaling weak   (default) every rank holds its own batch of the workload's size (c2: 100k molecules per GPU);
  --scaling strong ONE global set of 8 x <workload size> molecules (c4: 8 x 125k = 1 M, BASELINE configs[3]) partitioned
                   over the ranks with parallel.shard_by_edges; a rank walks its share in micro-batches of at most the
                   workload's size, accumulating gradients, then all-reduces once.

Prints ONE JSON line (rank 0) with the whole-job edges/s, the aggregator's live-measured roofline figures and, at
N=1, the CPU baseline (the oracle's dense restatement of the reference path) timed on this host's cores on a bounded
sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import subprocess
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
STRONG_CHUNKS = 8           # --scaling strong: the global set is 8 chunks of the workload's size
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TF = 157.3    # dense fp32 MFMA peak (MI355X_MICROARCH.md)
MFMA_16BIT_PEAK_TF = 2500.0  # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE.json configs[1] (the headline); c3/c4/c5 = configs[2..4] shapes")
    ap.add_argument("--mode", default="train", choices=["fwd", "train"],
                    help="train (default) = forward + backward + gradient all-reduce; fwd = inference pass only")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong = one global set of %d x the workload's molecules sharded over the ranks" % STRONG_CHUNKS)
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the CPU baseline leg (2/3 training, 1/3 forward)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements that start child processes (the fp32-pipe step time)")
    ap.add_argument("--side-workloads", default="auto", choices=["auto", "on", "off"],
                    help="the \"workloads\" side field (c1, c3, c4, c5 stepped in the same process after the headline): "
                         "auto = with the default headline (c2, train, weak, one GPU)")
    ap.add_argument("--side-scale", type=float, default=1.0, help="fraction of each side workload's molecules (tests)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ self-launch
def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks as a child torch.distributed.run and relay its JSON line.
    Nothing in this parent has touched the GPU (torch.cuda.device_count() does not initialise it on this image), and
    the child is a child process, never an exec.  With fewer GPUs than ranks the ranks share devices over gloo
    (a rehearsal of the N-rank code path, flagged in the JSON; its rate is not a scaling number)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus and "MPNN_DIST_BACKEND" not in env:
        env["MPNN_DIST_BACKEND"] = "gloo"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in lines[-1:]:
        print(ln, flush=True)
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout[-4000:])
        sys.exit(r.returncode or 1)
    sys.exit(0)


# ------------------------------------------------------------------------------------------------ measurement helpers
def pmc_traffic(workload, kernel_substr):
    """Per-launch HBM bytes of one kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
    separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes; tools/measure_all.sh +
    tools/pmc_summary.py are the only writers).  PMC collection needs the profiler, so it cannot happen inside this
    process; the figure is a property of (kernel, workload) and is reported with its source file, that file's hash and
    the commit it was measured at, so a stale figure is visible.  (None, None) when no pass exists."""
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(REPO, "profiles", "%s_pmc_%s.json" % (rnd, workload))
        try:
            with open(path, "rb") as f:
                raw = f.read()
            d = json.loads(raw)
            for k, v in d["kernels"].items():
                if kernel_substr in k and "bwd" not in k:
                    return v["hbm_bytes"], {"file": "profiles/" + os.path.basename(path),
                                            "sha256_16": hashlib.sha256(raw).hexdigest()[:16],
                                            "measured_at_commit": d.get("commit"), "kernel": k}
        except (OSError, ValueError, KeyError):
            continue
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
    if os.environ.get("MPNN_CPU_THREADS"):                 # explicit override only; default = every core this process may use
        n = min(n, int(os.environ["MPNN_CPU_THREADS"]))
    return n


def cpu_baseline_leg(mb, hidden, steps, mode, budget_s, first_batch=0):
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
    nmol = 0
    # the dense path holds a (B, N, N, mf, nf) tensor (plus its gradient and the pair messages when training):
    # keep one batch under ~16 GB by dropping the largest molecules of a batch (C5: 200 atoms at H = 256 is 10.5 GB
    # per molecule), and say so in `sample`
    budget_bytes = 16e9 / (3.0 if mode == "train" else 1.5)
    n_atoms = np.asarray(mb.n_atoms, dtype=np.int64)
    dropped = 0
    t0 = time.perf_counter()
    for b0 in range(first_batch * 16, mb.num_mols - 16, 16):
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
        nmol += len(ids)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": edges / dt, "unit": "edges/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample_molecules": nmol, "sample_seconds": dt,
            "extrapolated_seconds_for_2k_molecules": dt * 2000.0 / max(nmol, 1),
            "sample": "%d molecules (%d batches of 16, the reference's batch size) of the same synthetic set%s, dense "
                      "padded torch-CPU path, %s, %.1f s on %d threads (os.cpu_count() = %s) -- a time-budgeted sample; "
                      "SURVEY 8d's 2k-molecule subsample would take %.0f s at this rate"
                      % (nmol, nb, " (%d molecules too large for the dense path left out)" % dropped if dropped else "",
                         "forward+backward" if mode == "train" else "forward", dt, cores, os.cpu_count(),
                         dt * 2000.0 / max(nmol, 1))}


def cpu_baseline(mb, hidden, steps, mode, budget_s):
    """Both legs in one record: the headline field follows `mode`; the other leg rides along under "forward"/"train"."""
    fwd = cpu_baseline_leg(mb, hidden, steps, "fwd", budget_s / 3.0 if mode == "train" else budget_s)
    if mode != "train":
        return fwd
    out = cpu_baseline_leg(mb, hidden, steps, "train", budget_s * 2.0 / 3.0)
    out["forward"] = {k: fwd[k] for k in ("value", "unit", "sample_molecules", "sample_seconds",
                                          "extrapolated_seconds_for_2k_molecules")}
    return out


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


def segsum_calibration(ops, graph, F, dev):
    """The standalone segmented-sum aggregator (mpnn_segsum_f32: AdjMsgAgg on materialised message rows, the kernel the
    weighted / attention aggregators run) on this batch, outside the timed region: BASELINE's 'aggregator' line."""
    E, V = graph.num_edges, graph.num_nodes
    msg = torch.empty(E, F, device=dev).normal_()
    for _ in range(3):
        ops.segsum_raw(msg, graph.row_ptr, None, V, label="calib")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.segsum_raw(msg, graph.row_ptr, None, V, label="calib")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    alg = 4.0 * F * (E + V) + 4.0 * (V + 1)
    return {"kernel": "segsum_pair_kernel (mpnn_segsum_f32 on materialised (E, mf) message rows)", "bound": "hbm",
            "algorithmic_bytes_per_launch": alg, "formula": "4*mf*(E+V) + 4*(V+1)", "avg_launch_ms": ms,
            "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "timed": "10 launches after 3 warm-ups, outside the timed steps"}


def make_model(workload, hidden, T, dev):
    """The model of a workload: models.basic_model.BasicModel, or (c3 / c3a) models.att_model.BasicModel with the cheap
    readout (the metric times message + aggregate + update; the default Set2Vec readout -- 100 LSTM steps over every
    atom -- is outside it)."""
    from mpnn_amd.models.basic_model import BasicModel
    if workload in ("c3", "c3a"):
        from mpnn_amd.models.att_model import BasicModel as AttModel
        from mpnn_amd.mpnn_functions import AdjMsgAgg, AttMsgAgg, GraphLevelOutput
        return AttModel(hidden, 4, hidden, 1 if workload == "c3a" else 50, 8, message_opts={}, agg_opts={},
                        update_opts={}, readout_opts={}, message_steps=T, readout_func=GraphLevelOutput,
                        message_agg_func=AttMsgAgg if workload == "c3a" else AdjMsgAgg).to(dev)
    return BasicModel(hidden, 4, hidden, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                      message_steps=T).to(dev)


SIDE_KERNELS = ["message_aggregate", "segsum", "edge_message", "gru_update", "gru_update_bwd", "message_aggregate_bwd",
                "att_message_bwd"]


def side_workload(name, dev, steps=3, warmup=1, scale=1.0):
    """One of the other BASELINE configs in the same process, outside the headline's clock: `warmup` + `steps` forward
    passes, the same of training steps, then one pass with HIP events around the hot-path launches.  The molecules are
    the workload's own synthetic set (seed 317); the atom features are made on the device (synth.hashed_features: same
    distribution, no 5 GB host array at hidden 256) -- step times do not depend on their values."""
    from mpnn_amd import ops, parallel, synth
    from mpnn_amd.graph import MolGraph
    mols, hidden, T, dist_name, desc = WORKLOADS[name]
    mols = max(32, int(mols * scale))
    t_start = time.perf_counter()
    mb = synth.make_molecules(mols, hidden, seed=317, dist=dist_name, edge_features=4, atom_features=False)
    g = MolGraph.from_molbatch(mb, dev)
    g.prepare(tile_plan=(hidden == 64), wide_plan=(hidden in (128, 256)))
    V, E = g.num_nodes, g.num_edges
    afm = synth.hashed_features(torch.arange(V, device=dev), hidden)
    mask = torch.ones(V, 1, device=dev)
    torch.manual_seed(317)
    model = make_model(name, hidden, T, dev)
    bucket = parallel.GradientBucket([p for n, p in model.named_parameters() if not n.startswith("of.")])
    seed = torch.full((V, hidden), 1.0 / float(mols), device=dev)

    def fwd():
        with torch.no_grad():
            model.message_passing(afm, g, g, mask)

    def train():
        bucket.zero()
        state, _ = model.message_passing(afm, g, g, mask)
        state.backward(gradient=seed.view_as(state))

    def clock(step, n_warm, n):
        for _ in range(n_warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    fwd_ms = clock(fwd, warmup, steps)
    train_ms = clock(train, warmup, steps)
    timer = ops.KernelTimer(SIDE_KERNELS)
    ops.set_kernel_timer(timer)
    for _ in range(steps):
        train()
    torch.cuda.synchronize()
    ops.set_kernel_timer(None)
    per_launch = {k: timer.mean_ms(k) for k in sorted(timer.names) if timer.events[k]}
    per_step = {k: per_launch[k] * len(timer.events[k]) / steps for k in per_launch}
    dom = max(per_step, key=per_step.get) if per_step else None
    out = {"workload": "%s: %s" % (name, desc), "mols": mols, "atoms": V, "edges": E, "hidden": hidden, "mp_steps": T,
           "steps": steps, "warmup": warmup, "train_ms": train_ms, "fwd_ms": fwd_ms,
           "edges_per_s": E * T / (train_ms * 1e-3), "forward_edges_per_s": E * T / (fwd_ms * 1e-3),
           "kernels_ms_per_launch": per_launch, "kernels_ms_per_step": per_step,
           "dominant_kernel": ({"kernel": dom, "ms_per_launch": per_launch[dom], "ms_per_step": per_step[dom]} if dom else None)}
    if name == "c1":
        # the reference driver's own batching (test_lipo.py:150): batches of 16 molecules, each its own CSR
        parts = []
        for b0 in range(0, mols, 16):
            sub = synth.select(mb, np.arange(b0, min(b0 + 16, mols)))
            gs = MolGraph.from_molbatch(sub, dev)
            gs.prepare()
            a0, a1 = int(mb.atom_ptr[b0]), int(mb.atom_ptr[min(b0 + 16, mols)])
            parts.append((afm[a0:a1].contiguous(), gs, torch.ones(a1 - a0, 1, device=dev)))

        def epoch_fwd():
            with torch.no_grad():
                for a, gs, mk in parts:
                    model.message_passing(a, gs, gs, mk)

        def epoch_train():
            for a, gs, mk in parts:
                bucket.zero()
                state, _ = model.message_passing(a, gs, gs, mk)
                (state.sum() / 16).backward()

        f16, t16 = clock(epoch_fwd, warmup, steps), clock(epoch_train, warmup, steps)
        out["batches_of_16"] = {"batches": len(parts), "train_ms_per_epoch": t16, "fwd_ms_per_epoch": f16,
                                "train_edges_per_s": E * T / (t16 * 1e-3), "forward_edges_per_s": E * T / (f16 * 1e-3),
                                "note": "the same molecules as %d batches of 16 (the reference driver's batch size, "
                                        "test_lipo.py:150), one optimizer-sized step per batch" % len(parts)}
        # ... and the same epoch with every batch's step recorded into a HIP graph once and replayed (mpnn_amd/capture.py).
        # A child process: an illegal call inside a capture aborts the process, and a side figure must not take the headline down.
        try:
            r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "capture_step.py"), "--mols", str(mols), "--hidden",
                                str(hidden), "--steps", str(T), "--epochs", str(max(steps, 3))], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=300)
            out["batches_of_16"]["recorded_graphs"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:
            out["batches_of_16"]["recorded_graphs"] = {"error": repr(e)[:200]}
    out["wall_s"] = time.perf_counter() - t_start
    del model, bucket, afm, g, seed, mask
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                   # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU (or run without a launcher: "
                 "bench.py starts its own ranks)" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    ndev = max(torch.cuda.device_count(), 1)
    rehearsal = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MPNN_DIST_BACKEND", "nccl")      # "gloo": rehearsal with ranks sharing one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mpnn_amd import ops, parallel, synth
    from mpnn_amd.graph import MolGraph

    mols, hidden, T, dist_name, desc = WORKLOADS[args.workload]
    # ---- resident inputs: a list of micro-batches (afm, graph, mask); weak scaling has exactly one
    shard_info = None
    if args.scaling == "strong":
        micro, shard_info = parallel.strong_scaling_shard(STRONG_CHUNKS, mols, rank, world, seed=317, dist_name=dist_name,
                                                          micro_mols=mols)
        batches = []
        for mbi, keys in micro:
            g = MolGraph.from_molbatch(mbi, dev)
            a = synth.hashed_features(torch.from_numpy(keys).to(dev), hidden)
            batches.append((a, g, torch.ones(mbi.num_atoms, 1, device=dev)))
        mb = micro[0][0]
        local_mols = shard_info["local_mols"]
    else:
        mb = synth.make_molecules(mols, hidden, seed=317 + rank, dist=dist_name, edge_features=4)
        g = MolGraph.from_molbatch(mb, dev)
        batches = [(torch.from_numpy(mb.atom_feat).to(dev), g, torch.ones(mb.num_atoms, 1, device=dev))]
        local_mols = mols
    for _, g, _ in batches:                              # index arrays built once, outside the timed region
        g.prepare(tile_plan=(hidden == 64), wide_plan=(hidden in (128, 256)))
    afm, graph, mask = batches[0]
    V = sum(g.num_nodes for _, g, _ in batches)
    E = sum(g.num_edges for _, g, _ in batches)
    torch.manual_seed(317)                               # same weights on every rank
    model = make_model(args.workload, hidden, T, dev)
    hot = [p for n, p in model.named_parameters() if not n.startswith("of.")]   # readout is off the hot path
    bucket = parallel.GradientBucket(hot)
    total_mols = parallel.global_count(local_mols, dev)

    def step_fwd():
        with torch.no_grad():
            for a, g, mk in batches:
                state, _ = model.message_passing(a, g, g, mk)
        return state

    # d(loss)/d(state) of loss = sum(state) / total_mols (this shard's share of a global mean loss, i.e. the loss is
    # scaled by local_G / global_G), built once: what a readout's backward would hand to the path; seeding with it
    # skips a V x H reduce, expand and scale per step
    seeds = [torch.full((g.num_nodes, hidden), 1.0 / float(total_mols), device=dev) for _, g, _ in batches] \
        if args.mode == "train" else None
    ar_events = []

    def step_train():
        bucket.zero()
        for (a, g, mk), sd in zip(batches, seeds):
            state, _ = model.message_passing(a, g, g, mk)
            state.backward(gradient=sd.view_as(state))
        if ar_events is not None and dist is not None and backend == "nccl":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce()                          # ONE RCCL all-reduce of the flat gradient bucket
            e1.record()
            ar_events.append((e0, e1))
        else:
            t0 = time.perf_counter()
            bucket.all_reduce()
            if dist is not None:
                ar_events.append(time.perf_counter() - t0)
        return state

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        """W untimed steps, then exactly K steps between barrier+synchronize fences; max over ranks.  Nothing but
        the steps runs in the timed region (the per-kernel HIP events are a separate pass: kernel_pass)."""
        for _ in range(args.warmup):
            step()
        del ar_events[:]
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item())

    def kernel_pass(step, timer):
        """The same steps once more with two HIP events around every hot-path launch, on the launch stream
        (ops.KernelTimer): per-kernel average durations for the roofline lines, outside the headline's clock."""
        ops.set_kernel_timer(timer)
        saved = list(ar_events)
        for _ in range(min(args.steps, 5)):
            step()
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
        ar_events[:] = saved

    def gather_stat(x):
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = float(x)
        if dist is not None:
            dist.all_reduce(t)
        return t.tolist()

    edges_per_rank = gather_stat(E)
    mols_per_rank = gather_stat(local_mols)
    total_edges = float(sum(edges_per_rank))
    timer = ops.KernelTimer(SIDE_KERNELS)
    if args.mode == "train":
        dt_fwd = timed(step_fwd)
        dt = timed(step_train)
    else:
        dt_fwd = None
        dt = timed(step_fwd)
    ar_ms = None
    if ar_events:
        ar_ms = (sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events) if not isinstance(ar_events[0], float)
                 else 1e3 * sum(ar_events) / len(ar_events))
    kernel_pass(step_train if args.mode == "train" else step_fwd, timer)

    batch16 = None
    if args.workload == "c1" and world == 1:
        # the reference driver's own batching (test_lipo.py:150): 64 batches of 16 molecules per epoch, each with
        # its own CSR; launch-latency-bound on a GPU, reported beside the whole-set batch
        parts = []
        for b0 in range(0, mols, 16):
            sub = synth.select(mb, np.arange(b0, min(b0 + 16, mols)))
            gsub = MolGraph.from_molbatch(sub, dev)
            gsub.prepare()
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

        batch16 = (timed(epoch_fwd), timed(epoch_train))

    hoisted = None
    if world == 1 and hasattr(model, "hoist_message"):
        model.hoist_message = True
        hoisted = (timed(step_fwd), timed(step_train) if args.mode == "train" else float("nan"))
        model.hoist_message = False

    unfused_norm = None
    if world == 1 and getattr(model, "fuse_norm", False) and model._norm_fusable(afm):
        model.fuse_norm = False
        unfused_norm = (timed(step_fwd), timed(step_train) if args.mode == "train" else float("nan"))
        model.fuse_norm = True

    cold = None
    if world == 1 and args.scaling == "weak":
        # what a training loop that sees a NEW batch every step would pay on top of the resident-batch step time:
        # upload of the compact batch + CSR-derived index arrays, and the tile plan of the fused message+sum kernel
        def clock(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            return r, (time.perf_counter() - t0) * 1e3
        g2, up_ms = clock(lambda: MolGraph.from_molbatch(mb, dev))
        _, idx_ms = clock(lambda: g2.prepare(tile_plan=False))
        _, plan_ms = (clock(lambda: g2.tile_plan) if graph._tile_plan else
                      clock(lambda: g2.wide_plan) if graph._wide_plan else (None, 0.0))
        cold = {"upload_ms": up_ms, "index_arrays_ms": idx_ms, "tile_plan_ms": plan_ms,
                "total_ms": up_ms + idx_ms + plan_ms,
                "note": "per NEW batch, outside the timed steps (the headline keeps its batch resident in HBM): host -> "
                        "device copy of the compact batch, type order / transposed graph / destination list, and the "
                        "tile plan when the fused message+sum kernel runs"}
        del g2

    streaming = None
    if world == 1 and args.scaling == "weak" and args.mode == "train" and args.workload != "c1" and not args.no_side:
        # The same training step on a FRESH batch every step (as the reference's DataLoader loop, test_lipo.py:157-165): batch
        # t + 1 is uploaded, indexed and planned on a side stream by a worker thread while step t runs (mpnn_amd/streaming.py).
        # Three host batches of the workload's shape in rotation (structure only; every turn builds a new MolGraph from the
        # host arrays).  Atom features: made on the device from the atom index ("device_features"), or copied from pinned host
        # memory ("uploaded_features": 4 * H * V bytes more over PCIe per batch).
        from mpnn_amd.graph import MolGraph as _MG
        from mpnn_amd.streaming import BatchStream
        try:
            hosts = [synth.make_molecules(mols, hidden, seed=9000 + i, dist=dist_name, edge_features=4, atom_features=False)
                     for i in range(2)]
            hosts.append(synth.make_molecules(mols, hidden, seed=317, dist=dist_name, edge_features=4, atom_features=False))
            vmax = max(hb.num_atoms for hb in hosts)
            pool = torch.rand(vmax, hidden).mul_(2.0).sub_(1.0).pin_memory()
            n_steps = max(args.steps, 6)

            def run(upload):
                def make(hb, device):
                    g2 = _MG.from_molbatch(hb, device)
                    g2.prepare(tile_plan=(hidden == 64), wide_plan=(hidden in (128, 256)))
                    if upload:
                        f = pool[:g2.num_nodes].to(device, non_blocking=True)
                    else:
                        f = torch.empty(g2.num_nodes, hidden, device=device).uniform_(-1.0, 1.0)
                    return f, g2, torch.ones(g2.num_nodes, 1, device=device)
                bs = BatchStream((hosts[i % 3] for i in range(n_steps + 2)), dev, hidden, make_device_batch=make)
                t0, k, edges = None, 0, 0
                for b in bs:
                    if k == 2:                             # two warm-up steps (allocator, first launches)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    bucket.zero()
                    state, _ = model.message_passing(b.feats, b.graph, b.graph, b.mask)
                    state.backward(gradient=torch.full_like(state, 1.0 / float(total_mols)))
                    bs.done_with(b)
                    if k >= 2:
                        edges += b.graph.num_edges
                    k += 1
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t0
                return {"ms_per_step": dt_s / n_steps * 1e3, "edges_per_s": edges * T / dt_s}
            streaming = {"steps": n_steps, "device_features": run(False), "uploaded_features": run(True),
                         "resident_ms_per_step": dt / args.steps * 1e3,
                         "upload_bytes_per_batch_with_features": int(4 * hidden * vmax),
                         "note": "NOT the headline: every step takes a NEW batch (three host batches of the workload's shape in "
                                 "rotation, each turn re-uploaded and re-indexed: MolGraph.from_molbatch + prepare incl. the tile "
                                 "plan), prepared one batch ahead on a side stream by a worker thread (mpnn_amd/streaming.py)"}
            del pool, hosts
        except Exception as e:                            # a side figure must not take the headline down
            streaming = {"error": repr(e)[:300]}

    fp32_pipe = None
    if world == 1 and rank == 0 and not args.no_side and ops.math_mode() != "fp32":
        # the same step with every contraction on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, MPNN_GRU_MATH=fp32): what
        # the split arithmetic of the default path buys.  A child process (the switch is read once per process).
        torch.cuda.empty_cache()
        env = dict(os.environ, MPNN_GRU_MATH="fp32")
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", str(min(args.steps, 5)),
               "--warmup", "1", "--mode", args.mode, "--no-cpu", "--no-side"]
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            cj = json.loads(line)
            fp32_pipe = {"ms_per_step": cj["ms_per_step"], "value": cj["value"], "unit": cj["unit"],
                         "forward_ms_per_step": (cj.get("forward") or {}).get("ms_per_step"),
                         "kernels_ms": cj.get("kernels_ms"),
                         "note": "same workload and mode with MPNN_GRU_MATH=fp32 (strict fp32 MFMA, no operand splits), "
                                 "child process, %d steps" % min(args.steps, 5)}
        except Exception as e:                            # a side figure must not take the headline down
            fp32_pipe = {"error": repr(e)[:200]}

    if rank == 0:
        F = hidden
        nb = len(batches)
        # ---- the aggregator of the timed path.  Fused message+sum kernel when it ran, the standalone segmented sum
        # otherwise; per-launch figures are per micro-batch (this rank's E and V over its `nb` launches per step).
        Eb, Vb = E / nb, V / nb
        weighted = graph.agg_weight is not None
        fused_ms = timer.mean_ms("message_aggregate")
        if fused_ms is not None:
            # SURVEY 8(d) gives two aggregator byte formulas and asks to say which is used.  The fused kernel is priced
            # with the FUSED-GATHER one (a gathered source row per edge + the out rows + indices): that is the work the
            # reference's fused bmm does per step, and what a kernel without the LDS tile would have to move.  What this
            # kernel really has to fetch is less -- every h row once, every out row once, the plan words -- and is
            # reported beside it as `min_traffic`, the figure to hold PMC `traffic` against.
            alg_bytes = 4.0 * F * Eb + 4.0 * Eb + 4.0 * (Vb + 1) + 4.0 * F * Vb
            min_bytes = 4.0 * F * Vb + 4.0 * F * Vb + float(graph.plan_bytes())
            if F == 64:
                agg_ms, kname, ksub = fused_ms, "message_sum_tile_kernel (fused typed message + neighbour sum, mpnn_message_aggregate_f32)", "message_sum_tile"
            else:
                agg_ms, kname, ksub = fused_ms, "message_sum_wide_kernel (fused typed message + neighbour sum, aggregate-then-contract, mpnn_message_aggregate_wide_f32)", "message_sum_wide"
            formula = "SURVEY 8(d) fused-gather: 4*nf*E + 4*E + 4*(V+1) + 4*mf*V"
        else:
            alg_bytes = 4.0 * F * (Eb + Vb) + 4.0 * (Vb + 1) + (4.0 * Eb if weighted else 0.0)
            min_bytes = None
            agg_ms, kname, ksub = timer.mean_ms("segsum"), "segsum_pair_kernel (aggregator, mpnn_segsum_f32)", "segsum"
            formula = "SURVEY 8(d): 4*mf*(E+V) + 4*(V+1) (+4*E weights)"
        achieved = alg_bytes / (agg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload, ksub)
        out = {
            "metric": "edges/sec (message+aggregate+update)",
            "value": total_edges * T * args.steps / dt,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "mode": ("train: forward + backward + flat-gradient all-reduce" if args.mode == "train" else "forward only"),
                       "mols_per_gpu": local_mols, "atoms_per_gpu": V, "edges_per_gpu": E, "micro_batches_per_step": nb,
                       "hidden": hidden, "mp_steps": T,
                       "edge_features": 4, "edge_types": graph.num_types, "parallelism": "dp%d" % world,
                       "edges_counted": "directed edges x MP steps per pass",
                       "math": ops.math_description()},
            "roofline": {"kernel": kname, "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": agg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "formula": formula},
            "kernels_ms": {k: timer.mean_ms(k) for k in sorted(timer.names)},
        }
        if min_bytes is not None:
            out["roofline"]["frac_on_bytes_moved"] = min_bytes / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["min_traffic"] = {
                "formula": "4*nf*V + 4*mf*V + tile-plan words (h rows and out rows once each; gathers served from the LDS tile)",
                "bytes": min_bytes, "GB/s": min_bytes / (agg_ms * 1e-3) / 1e9, "frac_of_peak": min_bytes / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "the bytes this kernel must move; `traffic` (PMC) is to be read against this number"}
        if world > 1 or args.scaling == "strong":
            out["sharding"] = {"global_mols": int(sum(mols_per_rank)), "global_edges": int(total_edges),
                               "edges_per_rank_max": max(edges_per_rank), "edges_per_rank_min": min(edges_per_rank),
                               "mols_per_rank_max": max(mols_per_rank), "mols_per_rank_min": min(mols_per_rank),
                               "allreduce_ms": ar_ms, "allreduce_floats": int(bucket.flat.numel()),
                               "backend": backend or "none",
                               "partition": ("parallel.shard_by_edges over one global set" if args.scaling == "strong"
                                             else "independent equal batches per rank")}
        if rehearsal:
            out["rehearsal"] = ("%d ranks share %d GPU(s) over %s: exercises the multi-rank code path only, the rate is "
                                "not a scaling measurement" % (world, ndev, backend))
        # ---- the dense contractions against BOTH roofs: the matrix pipe (16-bit MFMA peak divided by the MFMAs issued per fp32
        # product: 6 for the three-way bf16 split, 3 for the two-way fp16 split; the fp32-MFMA-peak ratio is a side figure) and
        # HBM (the bytes the operator must move: operands read once, results written once).  `bound` names the roof the
        # kernel sits closer to; `frac` is the fraction of THAT roof.
        saved_rows = 4 if args.mode == "train" else 0         # the GRU forward dumps (r, z, n, gh_n) for the backward
        rows = []
        for key, name, flops, nbytes, what in (
                ("edge_message", "typed edge message (mpnn_edge_message_f32)", 2.0 * F * F * Eb, 4.0 * F * 2 * Eb,
                 "4*nf*E gathered + 4*mf*E written"),
                ("gru_update", "masked GRU update forward (mpnn_gru_update_f32)", 12.0 * F * F * Vb,
                 4.0 * F * (3 + saved_rows) * Vb, "m, h read, out%s written: 4*H*V each" % (" and 4 gate arrays" if saved_rows else "")),
                ("gru_update_bwd", "masked GRU update backward (mpnn_gru_update_bwd_f32)", 24.0 * F * F * Vb, 4.0 * F * 9 * Vb,
                 "dout, m, h, 4 gate arrays read, dm, dh written: 4*H*V each")):
            ms = timer.mean_ms(key)
            if ms is None:
                continue
            per = ops.mfma_per_product(key, hidden)
            peak_eq = MFMA_16BIT_PEAK_TF / per if per else MFMA_F32_PEAK_TF
            tf = flops / (ms * 1e-3) / 1e12
            gbs = nbytes / (ms * 1e-3) / 1e9
            f_mfma, f_hbm = tf / peak_eq, gbs / HBM_PEAK_GBS
            row = {"kernel": name, "avg_launch_ms": ms,
                   "mfma": {"unit": "TFLOP/s (fp32-equivalent)", "achieved": tf, "peak": peak_eq, "frac": f_mfma,
                            "peak_note": "%.0f TF 16-bit dense MFMA peak / %d MFMAs per fp32 product" % (MFMA_16BIT_PEAK_TF, per)
                                         if per else "fp32 MFMA peak",
                            "side_figure_vs_fp32_mfma_peak": tf / MFMA_F32_PEAK_TF},
                   "hbm": {"unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": f_hbm,
                           "algorithmic_bytes_per_launch": nbytes, "formula": what}}
            if f_hbm >= f_mfma:
                row.update({"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": f_hbm})
            else:
                row.update({"bound": "mfma", "unit": "TFLOP/s (fp32-equivalent)", "achieved": tf, "peak": peak_eq, "frac": f_mfma})
            rows.append(row)
        out["roofline_contractions"] = rows
        if world == 1:
            out["roofline"]["stream_calibration"] = stream_calibration(dev, alg_bytes)
            out["aggregator_standalone"] = segsum_calibration(ops, graph, F, dev)
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
        if cold is not None:
            out["cold_batch"] = cold
        if streaming is not None:
            out["streaming"] = streaming
        if fp32_pipe is not None:
            out["fp32_pipe"] = fp32_pipe
        if unfused_norm is not None:
            out["standalone_norm"] = {
                "train_ms_per_step": unfused_norm[1] / args.steps * 1e3, "forward_ms_per_step": unfused_norm[0] / args.steps * 1e3,
                "note": "NOT the headline: the same model with fuse_norm=False -- the masked norm after every update as "
                        "reduction + apply passes of its own (the headline takes the norm's moments in the update kernel's "
                        "epilogue, applies it where the next update reads its state, and runs the norms' backward inside the "
                        "GRU backward kernels: mpnn_gru_update_norm_f32 / mpnn_gru_update_norm_bwd_f32)"}
        if hoisted is not None:
            out["hoisted_message"] = {
                "train_ms_per_step": hoisted[1] / args.steps * 1e3, "forward_ms_per_step": hoisted[0] / args.steps * 1e3,
                "note": "NOT the headline: BasicModel.hoist_message=True computes message+aggregate once per pass "
                        "instead of once per MP step (their input is the constant afm, models/basic_model.py:57, so the "
                        "result is bit-identical); reported to show what the reference's own structure leaves on the table"}
        if world == 1 and not args.no_cpu:
            cmb = mb
            if args.scaling == "strong":                  # the strong-scaling shard carries no host features
                cmb = synth.make_molecules(min(mols, 4096), hidden, seed=317, dist=dist_name, edge_features=4)
            out["cpu_baseline"] = cpu_baseline(cmb, hidden, T, args.mode, args.cpu_seconds)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        side = args.side_workloads == "on" or (args.side_workloads == "auto" and args.workload == "c2" and args.mode == "train"
                                               and args.scaling == "weak" and not args.no_side)
        if world == 1 and side:
            # the other BASELINE configs, so that every config has a driver-run number: same process, after the headline's
            # timed region (the headline's batch stays resident: 288 GB hold all of them)
            out["workloads"] = {}
            for wl in ("c1", "c3", "c4", "c5"):
                try:
                    out["workloads"][wl] = side_workload(wl, dev, scale=args.side_scale)
                except Exception as e:                    # a side figure must not take the headline down
                    out["workloads"][wl] = {"error": repr(e)[:300]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
