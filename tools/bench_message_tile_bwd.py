"""A/B of the weight gradient of message+sum at the c2 size: the tile-plan kernel vs the per-edge gather kernel.
    python tools/bench_message_tile_bwd.py [n_mols]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import _lib, ops, synth             # noqa: E402
from mpnn_amd.graph import MolGraph                # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mb = synth.make_molecules(n, 64, seed=317)
g = MolGraph.from_molbatch(mb, dev).prepare()
h = torch.from_numpy(mb.atom_feat).to(dev)
K = g.num_types
dagg = torch.randn(g.num_nodes, 64, device=dev)
lib = _lib.load()
p = g.tile_plan


def tile():
    dA = torch.zeros(K, 64, 64, device=dev)
    _lib.check(lib.mpnn_message_aggregate_bwd_da_f32(_lib.fptr(dagg), _lib.fptr(h), _lib.iptr(p.tile_rec), _lib.iptr(p.tile_atom),
                                                      _lib.iptr(p.tile_rtk), _lib.iptr(p.slots), _lib.fptr(dA), g.num_nodes,
                                                      p.num_tiles, K, 64, 64, _lib.stream()), "tile bwd")
    return dA


def gather():
    dA = torch.zeros(K, 64, 64, device=dev)
    _lib.check(lib.mpnn_edge_message_agg_bwd_da_f32(_lib.fptr(dagg), _lib.fptr(h), _lib.iptr(g.col_idx), _lib.iptr(g.edge_dst), None,
                                                     _lib.iptr(g.order), _lib.iptr(g.type_ptr), None, _lib.fptr(dA), g.num_nodes,
                                                     g.num_edges, K, 64, 64, _lib.stream()), "gather bwd")
    return dA


for name, fn in (("tile-plan kernel", tile), ("per-edge gather kernel", gather)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(20):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-24s best %.3f ms (incl. the 64 KB zero fill)" % (name, best))
a, b = tile(), gather()
print("max |tile - gather| / max = %.2e" % float((a - b).abs().max() / b.abs().max()))
