#!/usr/bin/env python3
"""Reads the in-kernel cycle stamps of a -DMW_STAMP build of csrc/message_tile_wide.hip (diagnostic only):
    MPNN_EXTRA_HIPCC_FLAGS=-DMW_STAMP python -m mpnn_amd.build && python tools/stamp_message_wide.py [c4|c5]
then rebuild without the flag (the build notices the changed flag list).  Per phase of block 3, waves 0 and 7:
barrier wait | whole phase body | wait for the copies, and for active phases gather | guard+split | products."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import _lib, ops, synth  # noqa: E402
from mpnn_amd.graph import MolGraph  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "c4"
mols, F, dist = {"c4": (125_000, 128, "drug"), "c5": (50_000, 256, "skewed")}[w]
dev = torch.device("cuda:0")
mb = synth.make_molecules(mols, F, seed=317, dist=dist)
g = MolGraph.from_molbatch(mb, dev)
g.prepare(tile_plan=False, wide_plan=True)
h = torch.from_numpy(mb.atom_feat).to(dev)
A = torch.randn(g.num_types, F, F, device=dev) / F ** 0.5
lib = ctypes.CDLL(_lib.load()._name)
for _ in range(3):
    ops.message_aggregate_wide_raw(h, A, g)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib.mpnn_debug_mw_stamps(buf, 1)
ops.message_aggregate_wide_raw(h, A, g)
torch.cuda.synchronize()
lib.mpnn_debug_mw_stamps(buf, 0)
for wv, base in ((0, 0), (7, 16)):
    n = max(buf[base + 3], 1)
    na = max(buf[base + 7], 1)
    print("wave %d: %d phases: barrier %.0f  body %.0f  copy-wait %.0f cycles/phase | %d active: gather %.0f  guard+split %.0f  products %.0f"
          % (wv, buf[base + 3], buf[base] / n, buf[base + 1] / n, buf[base + 2] / n, buf[base + 7], buf[base + 4] / na,
             buf[base + 5] / na, buf[base + 6] / na))
