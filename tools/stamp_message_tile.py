"""Per-phase cycle sums of the fused message+sum tile kernel from in-kernel s_memtime stamps (diagnostic build only):
    MPNN_EXTRA_HIPCC_FLAGS=-DMT_STAMP python -m mpnn_amd.build --force && python tools/stamp_message_tile.py
then rebuild without the flag.  The stamps serialise what the real kernel overlaps: read the SHARES, not the total."""
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
for _ in range(3): ops.message_aggregate_tile_raw(h, A, g)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.mpnn_debug_mt_stamps(buf, 1)
ops.message_aggregate_tile_raw(h, A, g); torch.cuda.synchronize()
lib.mpnn_debug_mt_stamps(buf, 0)
for w, off in (("wave 0 (heavy block)", 0), ("wave 7 (light block)", 8)):
    n = buf[off + 7]
    print(w, "tiles stamped:", n)
    for name, v in zip(("stage_max (waits for the prefetched loads)", "barrier A", "stage_write", "barrier B", "issue next loads",
                        "row-tile loop", "out rows -> LDS -> HBM"), buf[off:off + 7]):
        print("   %-44s %8.1f cycles/tile" % (name, v / max(n, 1)))
    print("   total %.1f" % (sum(buf[off:off + 7]) / max(n, 1)))
