"""The tile plan of the fused message+sum kernel (graph.py::TilePlan), checked on the CPU: walking the plan exactly as
the kernel does (tiles -> sub-tiles -> row-tiles -> 16 slots whose ROW is a destination atom) must visit every edge once,
with the right source row, destination atom and bond type, and so reproduce sum_e A[type e] . h[src e] per atom."""
import numpy as np
import pytest
import torch

from mpnn_amd import synth
from mpnn_amd.graph import MolGraph


def _walk(g, h, A):
    p = g.tile_plan
    out = torch.zeros(g.num_nodes, h.shape[1], dtype=torch.float64)
    rec, ta, sl, se = (x.numpy() for x in (p.tile_rec, p.tile_atom, p.slots, p.slot_eid))
    seen = np.zeros(g.num_edges, int)
    for t in range(p.num_tiles):
        a0, n = rec[t, 0], rec[t, 1]
        assert 0 < n <= p.tile_atoms
        assert sorted(a for a in ta[t] if a >= 0) == list(range(a0, a0 + n))        # a permutation of the tile's atoms
        for B in range(8):
            last = -1
            for rt in range(rec[t, 2 + B], rec[t, 3 + B]):
                k = (sl[rt * 16] >> 16) & 15
                assert k >= last                               # a block's row-tiles: by type (then rank)
                last = k
                for m in range(16):
                    wd, e = sl[rt * 16 + m], se[rt * 16 + m]
                    assert ((wd >> 16) & 15) == k
                    if e < 0:
                        assert (wd & 0x7fff) == p.tile_atoms  # empty: not valid, reads the zero row
                        continue
                    assert (wd >> 14) & 1
                    d = ta[t, 16 * B + m]                      # row m of block B is this atom
                    assert d == g.edge_dst[e] and a0 + (wd & 0xff) == g.col_idx[e] and k == g.edge_type[e]
                    seen[e] += 1
                    out[d] += A[k] @ h[a0 + (wd & 0xff)]
    assert (seen == 1).all()
    # balance: blocks 2p / 2p+1 of a tile are its p-th heaviest / p-th lightest by row-tile count
    loads = np.diff(rec[:, 2:11], axis=1)
    assert (np.diff(loads[:, 0::2], axis=1) <= 0).all() and (np.diff(loads[:, 1::2], axis=1) >= 0).all()
    assert (loads[:, 0::2].min(axis=1) >= loads[:, 1::2].max(axis=1)).all()
    return out


@pytest.mark.parametrize("n_mols,seed,dist", [(300, 1, "drug"), (7, 2, "drug"), (1, 3, "drug"), (40, 4, "lipo")])
def test_plan_walk_reproduces_the_neighbour_sum(n_mols, seed, dist):
    mb = synth.make_molecules(n_mols, 8, seed=seed, dist=dist)
    g = MolGraph.from_molbatch(mb, torch.device("cpu"))
    p = g.tile_plan
    assert p is not None
    h = torch.randn(g.num_nodes, 8, dtype=torch.float64)
    A = torch.randn(g.num_types, 8, 8, dtype=torch.float64)
    ref = torch.zeros(g.num_nodes, 8, dtype=torch.float64)
    ref.index_add_(0, g.edge_dst.long(), torch.einsum("emn,en->em", A[g.edge_type.long()], h[g.col_idx.long()]))
    assert float((_walk(g, h, A) - ref).abs().max()) < 1e-12
    tp = p.tile_ptr.numpy()
    assert tp[0] == 0 and tp[-1] == g.num_nodes and set(tp.tolist()) <= set(g.graph_ptr.numpy().tolist())
    rec = p.tile_rec.numpy()
    assert np.array_equal(rec[:, 0], tp[:-1]) and np.array_equal(rec[:, 0] + rec[:, 1], tp[1:])
    assert np.array_equal(rec[:, 2:10].reshape(-1), p.rt_ptr.numpy()[:-1]) and np.array_equal(rec[:, 10], p.rt_ptr.numpy()[8::8])


def test_plan_edge_order_within_a_destination_and_type_is_rank_order():
    """Accumulation order = row-tile order, so rank r of (atom, type) must be that pair's r-th edge in CSR order."""
    mb = synth.make_molecules(50, 4, seed=8)
    g = MolGraph.from_molbatch(mb, torch.device("cpu"))
    p = g.tile_plan
    se = p.slot_eid.numpy().reshape(-1, 16)
    dst, typ = g.edge_dst.numpy(), g.edge_type.numpy()
    last = {}
    for rt in range(se.shape[0]):
        for e in se[rt]:
            if e >= 0:
                key = (dst[e], typ[e])
                assert last.get(key, -1) < e
                last[key] = e


def test_plan_refuses_what_the_kernel_cannot_do():
    big = synth.make_molecules(20, 4, seed=5, dist="skewed")          # molecules of up to 200 atoms: larger than a tile
    assert MolGraph.from_molbatch(big, torch.device("cpu")).tile_plan is None
    many = synth.make_molecules(20, 4, seed=6, edge_features=7)      # 7 bond types: more matrices than fit in LDS
    assert MolGraph.from_molbatch(many, torch.device("cpu")).tile_plan is None


def test_plan_refuses_an_edge_into_the_next_tile():
    """Two molecules of 100 atoms sit in two tiles (100 + 100 > 128).  One bond from the last atom of the first to the
    first atom of the second stays within 128 rows of the destination's tile start, yet leaves the tile: the plan must
    refuse (the ops then take the edge_message + segsum path) instead of gathering a row the tile does not hold."""
    import numpy as np
    n = 100
    src, dst = [], []
    for base in (0, n):
        for i in range(n - 1):                                  # a chain per molecule, both directions
            src += [base + i, base + i + 1]
            dst += [base + i + 1, base + i]
    def graph(extra):
        s, d = np.array(src + [e[0] for e in extra]), np.array(dst + [e[1] for e in extra])
        o = np.lexsort((s, d))
        s, d = s[o], d[o]
        row_ptr = np.zeros(2 * n + 1, np.int64)
        np.add.at(row_ptr, d + 1, 1)
        row_ptr = np.cumsum(row_ptr)
        t = lambda a, dt: torch.from_numpy(np.asarray(a)).to(dt)
        return MolGraph(t(row_ptr, torch.int32), t(s, torch.int32), None, torch.zeros(len(s), dtype=torch.int32),
                        torch.eye(1, 4), t([0, n, 2 * n], torch.int32))
    assert graph([]).tile_plan is not None
    # destination = last atom of tile 0, source = atom 105: 105 - 0 < 128 passes a start-relative bound
    assert graph([(n + 5, n - 1), (n - 1, n + 5)]).tile_plan is None


# ------------------------------------------------------------------------------------------ wide plan (widths 128 / 256)
def _walk_wide(g, h, A):
    """Walk graph.WidePlan exactly as csrc/message_tile_wide.hip does: per tile, per block of 32 sorted atoms, per bond
    type: S = per-atom sum of the slot rows' source rows, then out[atom] += A_k . S."""
    p = g.wide_plan
    K = g.num_types
    out = torch.zeros(g.num_nodes, h.shape[1], dtype=torch.float64)
    rec, ta, bo = p.tile_rec.numpy(), p.tile_atom.numpy(), p.blk_off.numpy()
    sl, se = p.slots.numpy().reshape(-1, 32), p.slot_eid.numpy().reshape(-1, 32)
    seen = np.zeros(g.num_edges, int)
    dst, typ = g.edge_dst.numpy(), g.edge_type.numpy()
    for t in range(p.num_tiles):
        a0, n, row0, nrows = rec[t]
        assert 0 < n <= p.TILE_ATOMS and nrows <= p.MAX_ROWS and bo[t, 0] == 0 and bo[t, -1] == nrows
        assert sorted(a for a in ta[t] if a >= 0) == list(range(a0, a0 + n))
        for b in range(8):
            for k in range(K):
                lo, hi = bo[t, b * K + k], bo[t, b * K + k + 1]
                S = torch.zeros(32, h.shape[1], dtype=torch.float64)
                for rr in range(row0 + lo, row0 + hi):
                    last = {}
                    for m in range(32):
                        w, e = sl[rr, m], se[rr, m]
                        if e < 0:
                            assert w == p.TILE_ATOMS
                            continue
                        d = ta[t, 32 * b + m]
                        assert d == dst[e] and a0 + w == g.col_idx[e] and typ[e] == k
                        seen[e] += 1
                        S[m] += h[a0 + w]
                for m in range(32):
                    if ta[t, 32 * b + m] >= 0 and hi > lo:
                        out[ta[t, 32 * b + m]] += A[k] @ S[m]
    assert (seen == 1).all()
    return out


@pytest.mark.parametrize("n_mols,seed,dist", [(120, 1, "drug"), (3, 2, "drug"), (1, 3, "drug"), (30, 4, "skewed")])
def test_wide_plan_walk_reproduces_the_neighbour_sum(n_mols, seed, dist):
    mb = synth.make_molecules(n_mols, 8, seed=seed, dist=dist)
    g = MolGraph.from_molbatch(mb, torch.device("cpu"))
    assert g.wide_plan is not None                          # molecules of up to 200 atoms fit a 256-atom tile
    h = torch.randn(g.num_nodes, 8, dtype=torch.float64)
    A = torch.randn(g.num_types, 8, 8, dtype=torch.float64)
    ref = torch.zeros(g.num_nodes, 8, dtype=torch.float64)
    ref.index_add_(0, g.edge_dst.long(), torch.einsum("emn,en->em", A[g.edge_type.long()], h[g.col_idx.long()]))
    assert float((_walk_wide(g, h, A) - ref).abs().max()) < 1e-12


def test_wide_plan_rank_order_is_edge_order_and_limits():
    mb = synth.make_molecules(40, 4, seed=8, dist="skewed")
    g = MolGraph.from_molbatch(mb, torch.device("cpu"))
    p = g.wide_plan
    se = p.slot_eid.numpy().reshape(-1, 32)
    last = {}
    dst, typ = g.edge_dst.numpy(), g.edge_type.numpy()
    for rr in range(se.shape[0]):
        for e in se[rr]:
            if e >= 0:
                key = (dst[e], typ[e])
                assert last.get(key, -1) < e                 # rank r of (atom, type) = its r-th edge in CSR order
                last[key] = e
    many = synth.make_molecules(20, 4, seed=6, edge_features=9)      # 9 bond types: more than the kernel's phases take
    assert MolGraph.from_molbatch(many, torch.device("cpu")).wide_plan is None


def _check_wide_fast(g):
    """Vectorised form of _walk_wide's structural assertions (every tile's atom list a permutation of the tile, every edge
    in exactly one slot, of its own destination row, source and type) for batches too large for the Python walk."""
    p = g.wide_plan
    assert p is not None
    K = g.num_types
    rec, ta, bo = p.tile_rec.numpy().astype(np.int64), p.tile_atom.numpy().astype(np.int64), p.blk_off.numpy().astype(np.int64)
    sl, se = p.slots.numpy().astype(np.int64).reshape(-1, 32), p.slot_eid.numpy().astype(np.int64).reshape(-1, 32)
    for t in range(p.num_tiles):
        a = np.sort(ta[t][ta[t] >= 0])
        assert np.array_equal(a, np.arange(rec[t, 0], rec[t, 0] + rec[t, 1]))
    placed = se[se >= 0]
    assert len(placed) == g.num_edges and np.array_equal(np.sort(placed), np.arange(g.num_edges))
    # slot row rr belongs to (tile, block, type): recover them from blk_off
    row_tile = np.repeat(np.arange(p.num_tiles), rec[:, 3])
    assert len(row_tile) == sl.shape[0]
    local = np.arange(sl.shape[0]) - rec[row_tile, 2]
    grp = np.array([np.searchsorted(bo[t, 1:], l, side="right") for t, l in zip(row_tile, local)])
    blk, typ = grp // K, grp % K
    rr, m = np.nonzero(se >= 0)
    e = se[rr, m]
    dst, src, et = g.edge_dst.numpy(), g.col_idx.numpy(), g.edge_type.numpy()
    assert np.array_equal(ta[row_tile[rr], 32 * blk[rr] + m], dst[e])
    assert np.array_equal(rec[row_tile[rr], 0] + sl[rr, m], src[e]) and np.array_equal(typ[rr], et[e])
    assert (sl[se < 0] == p.TILE_ATOMS).all()


@pytest.mark.parametrize("ef,n_mols", [(7, 5000), (8, 5000), (8, 60)])
def test_wide_plan_at_seven_and_eight_bond_types(ef, n_mols):
    """the sort key used to pack (tile, per-type counts) into one int64 -- tile * 256 ** 7 wraps beyond
    128 tiles and 256 ** 8 does not fit at all.  5000 molecules = ~147 tiles."""
    mb = synth.make_molecules(n_mols, 8, seed=11, edge_features=ef)
    g = MolGraph.from_molbatch(mb, torch.device("cpu"))
    assert g.num_types == ef
    assert g.wide_plan is not None and (n_mols < 5000 or g.wide_plan.num_tiles > 128)
    _check_wide_fast(g)
    if n_mols <= 60:
        h = torch.randn(g.num_nodes, 8, dtype=torch.float64)
        A = torch.randn(g.num_types, 8, 8, dtype=torch.float64)
        ref = torch.zeros(g.num_nodes, 8, dtype=torch.float64)
        ref.index_add_(0, g.edge_dst.long(), torch.einsum("emn,en->em", A[g.edge_type.long()], h[g.col_idx.long()]))
        assert float((_walk_wide(g, h, A) - ref).abs().max()) < 1e-12
