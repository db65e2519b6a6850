import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops, synth, _lib
from mpnn_amd.graph import MolGraph
dev = torch.device("cuda:0")
mb = synth.make_molecules(100_000, 64, seed=317)
g = MolGraph.from_molbatch(mb, dev).prepare()
h = torch.from_numpy(mb.atom_feat).to(dev)
A = torch.randn(g.num_types, 64, 64, device=dev) / 8.0
lib = _lib.load()
for _ in range(3): ops.message_aggregate_tile_raw(h, A, None, g)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.mpnn_debug_mt_stamps(buf, 1)
ops.message_aggregate_tile_raw(h, A, None, g); torch.cuda.synchronize()
lib.mpnn_debug_mt_stamps(buf, 0)
n = buf[5]
print("row-tiles stamped:", n)
for name, v in zip(("a prefetch issue", "b first MFMAs", "c next matrix", "d combine", "e pair epilogue+rotate"), buf[:5]):
    print("%-24s %8.1f cycles/row-tile" % (name, v / max(n, 1)))
print("total %.1f" % (sum(buf[:5]) / max(n, 1)))
