#!/usr/bin/env python3
"""The standalone segmented-sum aggregator (mpnn_segsum_f32) on a synthetic graph, beside plain copy / add streams of
the same byte count: ms and algorithmic GB/s.   python tools/bench_segsum.py [F] [drug|skewed] [mols]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops, synth  # noqa: E402
from mpnn_amd.graph import MolGraph  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dist = sys.argv[2] if len(sys.argv) > 2 else "drug"
mols = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
dev = torch.device("cuda:0")
mb = synth.make_molecules(mols, 4, seed=317, dist=dist)
g = MolGraph.from_molbatch(mb, dev)
E, V = g.num_edges, g.num_nodes
msg = torch.randn(E, F, device=dev)
w = torch.rand(E, device=dev)
for weights in (None, w):
    for _ in range(5):
        ops.segsum_raw(msg, g.row_ptr, weights, V)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    a.record()
    for _ in range(n):
        ops.segsum_raw(msg, g.row_ptr, weights, V)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    by = 4.0 * F * (E + V) + 4.0 * (V + 1) + (4.0 * E if weights is not None else 0)
    print("variant=%s F=%d dist=%s weights=%s  %.4f ms  %.0f GB/s" % ("pair+nt", F, dist,
          weights is not None, ms, by / ms / 1e6))

# calibration: plain streaming kernels of the same read:write mix (torch elementwise), same bytes
n = (E + V) * F // 3
x, y, z = (torch.randn(n, device=dev) for _ in range(3))
for name, fn, nbytes in (("copy (1R:1W)", lambda: z.copy_(x), 8.0 * n), ("add (2R:1W)", lambda: torch.add(x, y, out=z), 12.0 * n)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    print("calibration %-14s %.4f ms  %.0f GB/s" % (name, ms, nbytes / ms / 1e6))
