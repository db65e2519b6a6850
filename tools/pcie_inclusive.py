#!/usr/bin/env python3
"""The reference's wire format is a dense padded HOST batch (afm, bfm, adj, mask; pre_process/data_loader.py:50-70).
This times what it costs to enter the hot path that way -- host-to-device copy of the dense tensors, dense -> CSR on the
device (mpnn_csr_count / mpnn_csr_fill), then one training pass -- next to the resident-batch rate bench.py reports.

    python tools/pcie_inclusive.py [molecules]        (default 10000, hidden 64, c2-shaped molecules)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import synth                                   # noqa: E402
from mpnn_amd.graph import MolGraph                          # noqa: E402
from mpnn_amd.models.basic_model import BasicModel           # noqa: E402


def main():
    mols = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    H, T = 64, 3
    dev = torch.device("cuda:0")
    mb = synth.make_molecules(mols, H, seed=317)
    host = {k: torch.from_numpy(v).pin_memory() for k, v in synth.to_dense(mb).items()}
    nbytes = sum(v.numel() * v.element_size() for v in host.values())
    torch.manual_seed(317)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                       message_steps=T).to(dev)

    def sync():
        torch.cuda.synchronize()

    def one(batch_on_host):
        sync(); t0 = time.perf_counter()
        d = {k: v.to(dev, non_blocking=True) for k, v in batch_on_host.items()}
        sync(); t1 = time.perf_counter()
        g = MolGraph.from_dense(d["adj"], d["bfm"])
        g.order, g.type_ptr, g.transpose, g.edge_dst
        sync(); t2 = time.perf_counter()
        for p in model.parameters():
            p.grad = None
        state, _ = model.message_passing(d["afm"], g, g, d["mask"])
        state.sum().backward()
        sync(); t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2, g.num_edges

    for _ in range(2):
        one(host)
    reps = [one(host) for _ in range(5)]
    h2d, csr, step = (min(r[i] for r in reps) for i in range(3))
    E = reps[0][3]
    print("%d molecules, dense host batch %.2f GB (pinned), E = %d directed edges" % (mols, nbytes / 1e9, E))
    print("H2D copy %.1f ms (%.1f GB/s)   dense->CSR + index arrays %.1f ms   training pass %.1f ms"
          % (h2d * 1e3, nbytes / h2d / 1e9, csr * 1e3, step * 1e3))
    print("edges x %d steps / s:  resident %.3f G   incl. CSR build %.3f G   incl. H2D + CSR build %.3f G"
          % (T, T * E / step / 1e9, T * E / (step + csr) / 1e9, T * E / (step + csr + h2d) / 1e9))


if __name__ == "__main__":
    main()
