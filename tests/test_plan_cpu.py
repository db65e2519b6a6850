"""The tile plan of the fused message+sum kernels (graph.py::TilePlan), checked on the CPU: walking the plan exactly as
the kernel does (tiles -> sub-tiles -> row-tiles -> 16 slots) must visit every edge once, with the right source row,
destination row and bond type, and so reproduce sum_e A[type e] . h[src e] per destination atom."""
import numpy as np
import pytest
import torch

from mpnn_amd import synth
from mpnn_amd.graph import MolGraph


def _walk(g, h, A):
    p = g.tile_plan
    out = torch.zeros(g.num_nodes, h.shape[1], dtype=torch.float64)
    tp, rp, rty, sl, se = (x.numpy() for x in (p.tile_ptr, p.rt_ptr, p.rt_type, p.slots, p.slot_eid))
    seen = np.zeros(g.num_edges, int)
    for t in range(p.num_tiles):
        a0, n = tp[t], tp[t + 1] - tp[t]
        assert 0 < n <= p.tile_atoms
        ss = (n + 3) // 4
        for q in range(4):
            last_type = -1
            for rt in range(rp[4 * t + q], rp[4 * t + q + 1]):
                k = rty[rt]
                assert k >= last_type                         # a sub-tile's row-tiles are grouped by type
                last_type = k
                for s in range(16):
                    wd, e = sl[rt * 16 + s], se[rt * 16 + s]
                    if e < 0:
                        assert wd == 32 << 8                  # padding: source row 0, sink destination row
                        continue
                    src, d = a0 + (wd & 0xff), a0 + q * ss + ((wd >> 8) & 0x3f)
                    assert src == g.col_idx[e] and d == g.edge_dst[e] and k == g.edge_type[e]
                    assert ((wd >> 8) & 0x3f) < min(ss, 32)
                    seen[e] += 1
                    out[d] += A[k] @ h[src]
    assert (seen == 1).all()
    return out


@pytest.mark.parametrize("n_mols,seed,dist", [(300, 1, "drug"), (7, 2, "drug"), (1, 3, "drug"), (40, 4, "lipo")])
def test_plan_walk_reproduces_the_neighbour_sum(n_mols, seed, dist):
    mb = synth.make_molecules(n_mols, 8, seed=seed, dist=dist)
    g = MolGraph.from_molbatch(mb, torch.device("cpu"))
    assert g.tile_plan is not None
    h = torch.randn(g.num_nodes, 8, dtype=torch.float64)
    A = torch.randn(g.num_types, 8, 8, dtype=torch.float64)
    ref = torch.zeros(g.num_nodes, 8, dtype=torch.float64)
    ref.index_add_(0, g.edge_dst.long(), torch.einsum("emn,en->em", A[g.edge_type.long()], h[g.col_idx.long()]))
    assert float((_walk(g, h, A) - ref).abs().max()) < 1e-12
    tp = g.tile_plan.tile_ptr.numpy()
    assert tp[0] == 0 and tp[-1] == g.num_nodes and set(tp.tolist()) <= set(g.graph_ptr.numpy().tolist())


def test_plan_refuses_what_the_kernel_cannot_do():
    big = synth.make_molecules(20, 4, seed=5, dist="skewed")          # molecules of up to 200 atoms: larger than a tile
    assert MolGraph.from_molbatch(big, torch.device("cpu")).tile_plan is None
    many = synth.make_molecules(20, 4, seed=6, edge_features=7)      # 7 bond types: more matrices than fit in LDS
    assert MolGraph.from_molbatch(many, torch.device("cpu")).tile_plan is None
