#!/usr/bin/env python3
"""In-kernel cycle stamps of the gated message + sum backward (csrc/message_tile_wide.hip, att_message_bwd_tile_kernel) at
c3's size; diagnostic build in its own library directory:
    MPNN_EXTRA_HIPCC_FLAGS=-DMB_STAMP python -m mpnn_amd.build && MPNN_EXTRA_HIPCC_FLAGS=-DMB_STAMP python tools/stamp_att_bwd.py
Without the flag the script only times the two backward kernels (HIP events)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import _lib, ops, synth  # noqa: E402
from mpnn_amd.graph import MolGraph  # noqa: E402

mols = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
mb = synth.make_molecules(mols, 128, seed=317, atom_features=False)
g = MolGraph.from_molbatch(mb, dev).prepare(tile_plan=False, wide_plan=True)
V, K, F = g.num_nodes, g.num_types, 128
gen = torch.Generator(device=dev).manual_seed(0)
h, z, cot = (torch.randn(V, F, device=dev, generator=gen) for _ in range(3))
A = torch.randn(K, F, F, device=dev, generator=gen) / F ** 0.5
q = torch.randn(K, F, device=dev, generator=gen)
out, ws = ops.message_aggregate_wide_gated_raw(h, A, z, q, g, keep_workspace=True)
NAMES = ["att_message_bwd", "message_aggregate_bwd"]
timer = ops.KernelTimer(NAMES)
for it in range(4):
    if it == 1:
        ops.set_kernel_timer(timer)
    ops.message_aggregate_wide_gated_bwd_raw(h, A, z, q, cot, ws, g)
torch.cuda.synchronize()
ops.set_kernel_timer(None)
print("V = %d atoms, E = %d edges; kernel ms:" % (V, g.num_edges), {k: round(timer.mean_ms(k), 3) for k in NAMES})
lib = ctypes.CDLL(_lib.load()._name)
if hasattr(lib, "mpnn_debug_mb_stamps"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.mpnn_debug_mb_stamps(buf, 1)
    ops.message_aggregate_wide_gated_bwd_raw(h, A, z, q, cot, ws, g)
    torch.cuda.synchronize()
    lib.mpnn_debug_mb_stamps(buf, 0)
    tiles, work = max(buf[15], 1), max(buf[14], 1)
    ph = 4 * K * tiles
    print("block 3 wave 0: %d tiles, %d phases with work of %d" % (tiles, work, ph))
    print("per tile: prologue (dout rows, split) %.0f | finishing %.0f" % (buf[0] / tiles, buf[8] / tiles))
    print("per phase: barrier %.0f | copy issue %.0f | chunk-end store %.0f | copy wait %.0f" %
          (buf[1] / ph, buf[2] / ph, buf[6] / ph, buf[7] / ph))
    print("per phase with work: 24 MFMAs issued %.0f | gather %.0f | gate, u, sums %.0f" % (buf[3] / work, buf[4] / work, buf[5] / work))
    print("finishing: store wait %.0f | barrier %.0f | dout requests %.0f | publish + row loop %.0f" %
          tuple(buf[i] / tiles for i in (9, 10, 11, 12)))
    tot = sum(buf[i] for i in range(9))
    print("sum per tile %.0f cycles" % (tot / tiles))
