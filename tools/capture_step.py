#!/usr/bin/env python3
"""configs[0] in the reference driver's own batching (test_lipo.py:150: batches of 16 molecules), stepped eagerly and as
recorded HIP graphs (mpnn_amd/capture.py), one graph per batch, replayed every epoch.

    python tools/capture_step.py [--mols 1024] [--hidden 22] [--epochs 5] [--unsafe]

Prints one JSON line: edges/s of the eager epoch and of the replayed one, launches per batch, and the largest difference
between the gradients of the two.  bench.py runs it as a CHILD process for its "workloads" side field: an illegal call inside
a capture aborts the process without raising (the core dumps of rounds 2 and 3, gpurun_out/gcap.log and b_c1_g.err, were
this script's predecessor recording a model whose AccumulateGrad nodes lived on the default stream -- see capture.py);
--unsafe reproduces that state on purpose: it skips drop_cached_autograd_state and warms up on the default stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=22)
    ap.add_argument("--steps", type=int, default=3, help="message-passing steps of the model")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--unsafe", action="store_true")
    args = ap.parse_args()
    from mpnn_amd import capture, parallel, synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel
    dev = torch.device("cuda:0")
    H, T, B = args.hidden, args.steps, args.batch
    mb = synth.make_molecules(args.mols, H, seed=317, dist="lipo", edge_features=4)
    torch.manual_seed(317)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T).to(dev)
    bucket = parallel.GradientBucket([p for n, p in model.named_parameters() if not n.startswith("of.")])
    parts = []
    for b0 in range(0, args.mols, B):
        sub = synth.select(mb, np.arange(b0, min(b0 + B, args.mols)))
        gs = MolGraph.from_molbatch(sub, dev).prepare(tile_plan=(H == 64), wide_plan=(H in (128, 256)))
        a = torch.from_numpy(sub.atom_feat).to(dev)
        parts.append((a, gs, torch.ones(a.shape[0], 1, device=dev), torch.full((a.shape[0], H), 1.0 / B, device=dev)))

    def eager(a, gs, mk, sd):
        bucket.zero()
        state, _ = model.message_passing(a, gs, gs, mk)
        state.backward(gradient=sd)

    def clock(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_eager = clock(lambda: [eager(*p) for p in parts], args.epochs)
    want = []
    for p in parts[:4]:
        eager(*p)
        want.append(bucket.flat.clone())
    t0 = time.perf_counter()
    if args.unsafe:
        # what the aborted attempts did: the model above has run on the default stream and still holds edge_embed
        caps = []
        for a, gs, mk, sd in parts:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eager(a, gs, mk, sd)
            caps.append(g)
        replay = lambda: [g.replay() for g in caps]
    else:
        caps = [capture.capture_training_step(model, a, gs, mk, sd, bucket) for a, gs, mk, sd in parts]
        replay = lambda: [c.replay() for c in caps]
    torch.cuda.synchronize()
    t_record = time.perf_counter() - t0
    worst = 0.0
    for c, w in zip(caps[:4], want):
        c.replay()
        torch.cuda.synchronize()
        worst = max(worst, float((bucket.flat - w).abs().max()) / max(float(w.abs().max()), 1e-30))
    t_graph = clock(replay, args.epochs)
    edges = int(mb.num_edges) * T
    print(json.dumps({"mols": args.mols, "hidden": H, "mp_steps": T, "batch": B, "batches": len(parts), "edges": int(mb.num_edges),
                      "eager_train_ms_per_epoch": t_eager * 1e3, "eager_train_edges_per_s": edges / t_eager,
                      "recorded_train_ms_per_epoch": t_graph * 1e3, "recorded_train_edges_per_s": edges / t_graph,
                      "recording_s": t_record, "max_rel_gradient_difference_recorded_vs_eager": worst,
                      "note": "one HIP graph per batch of %d molecules (forward + backward of the whole step), recorded once "
                              "and replayed every epoch; the optimizer step is outside both clocks" % B}), flush=True)


if __name__ == "__main__":
    main()
