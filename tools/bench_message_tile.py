"""A/B of message+sum at the c2 size: the fused tile kernel vs the two-kernel path (message rows to HBM + segmented sum).
    python tools/bench_message_tile.py [n_mols]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops, synth                     # noqa: E402
from mpnn_amd.graph import MolGraph                 # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mb = synth.make_molecules(n, 64, seed=317)
g = MolGraph.from_molbatch(mb, dev).prepare()
h = torch.from_numpy(mb.atom_feat).to(dev)
A = torch.randn(g.num_types, 64, 64, device=dev) / 8.0
V, E = g.num_nodes, g.num_edges
p = g.tile_plan
print("V %d E %d tiles %d row-tiles %d slot fill %.3f plan bytes %.1f MB" % (V, E, p.num_tiles, p.num_row_tiles,
                                                                              E / (16.0 * p.num_row_tiles), p.nbytes / 1e6))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best, tot = 1e9, 0.0
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = min(best, ms)
        tot += ms
    return best, tot / reps


for name, fn in (("tile kernel", lambda: ops.message_aggregate_tile_raw(h, A, g)),
                 ("message + segsum", lambda: ops.segsum_raw(ops.edge_message_raw(h, A, g), g.row_ptr, None, V))):
    best, avg = timeit(fn)
    byt = 8.0 * 64 * V + p.nbytes
    print("%-18s best %.3f ms avg %.3f ms   (min-traffic %.2f GB -> %.0f GB/s at best)" % (name, best, avg, byt / 1e9,
                                                                                            byt / best / 1e6))
a = ops.message_aggregate_tile_raw(h, A, g)
b = ops.segsum_raw(ops.edge_message_raw(h, A, g), g.row_ptr, None, V)
print("max |tile - two-kernel| = %.3e" % float((a - b).abs().max()))
