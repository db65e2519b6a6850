#!/usr/bin/env python3
"""Fused message+sum at widths 128 / 256 (mpnn_message_aggregate_wide_f32) against the two kernels it replaces
(mpnn_edge_message_f32 + mpnn_segsum_f32), per launch, on the c4 / c5 shapes:  python tools/bench_message_wide.py [c4|c5|c2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops, synth  # noqa: E402
from mpnn_amd.graph import MolGraph  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    w = sys.argv[1] if len(sys.argv) > 1 else "c4"
    mols, F, dist = {"c4": (125_000, 128, "drug"), "c5": (50_000, 256, "skewed"), "c2": (100_000, 64, "drug")}[w]
    dev = torch.device("cuda:0")
    mb = synth.make_molecules(mols, F, seed=317, dist=dist)
    g = MolGraph.from_molbatch(mb, dev)
    g.prepare(tile_plan=False, wide_plan=True)
    h = torch.from_numpy(mb.atom_feat).to(dev)
    A = torch.randn(g.num_types, F, F, device=dev) / F ** 0.5
    V, E = g.num_nodes, g.num_edges
    p = g.wide_plan
    t_f = timeit(lambda: ops.message_aggregate_wide_raw(h, A, g))
    t_m = timeit(lambda: ops.edge_message_raw(h, A, g))
    msg = ops.edge_message_raw(h, A, g)
    t_s = timeit(lambda: ops.segsum_raw(msg, g.row_ptr, None, V))
    moved = 8.0 * F * V + p.nbytes
    ref = ops.segsum_raw(msg, g.row_ptr, None, V)
    err = float((ops.message_aggregate_wide_raw(h, A, g) - ref).abs().max())
    print("%s: V %d E %d F %d tiles %d slot rows %d (fill %.2f)" % (w, V, E, F, p.num_tiles, p.num_rows, E / (32.0 * p.num_rows)))
    print("  fused %.3f ms = %.2f TB/s on the %.2f GB it must move   |   message %.3f + segsum %.3f = %.3f ms   |   max diff %.2e"
          % (t_f, moved / t_f / 1e9, moved / 1e9, t_m, t_s, t_m + t_s, err))


if __name__ == "__main__":
    main()
