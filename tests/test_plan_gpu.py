"""The per-batch index kernels (csrc/plan.hip) against the torch builders of mpnn_amd/graph.py, bit for bit: destination list,
type order, transposed graph, TilePlan (width 64) and WidePlan (widths 128 / 256) -- small and ragged batches, every bond-type
count the plans take, skewed molecules of up to 200 atoms, batches the plans must refuse."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pair(dev, mb, monkeypatch):
    """(graph built by the kernels, graph built by the torch ops)"""
    from mpnn_amd import graph as G
    fast = G.MolGraph.from_molbatch(mb, dev)
    slow = G.MolGraph.from_molbatch(mb, dev)
    calls, count = [], G._plan_count
    with monkeypatch.context() as m:
        m.setattr(G, "_plan_count", lambda *a, **k: (calls.append(1), count(*a, **k))[1])
        fast.prepare(tile_plan=True, wide_plan=True)
    assert len(calls) >= 1, "the plans of the first graph must come from the index kernels"   # 1: a molecule over 128 atoms
    with monkeypatch.context() as m:
        m.setattr(G, "_fast_path", lambda *a, **k: False)
        slow.prepare(tile_plan=True, wide_plan=True)
    return fast, slow


def _same(a, b, what):
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    assert torch.equal(a, b), what


@pytest.mark.parametrize("n_mols,seed,dist,K", [(1, 1, "drug", 4), (7, 2, "drug", 4), (300, 3, "drug", 4), (5000, 4, "drug", 4),
                                                (40, 5, "lipo", 4), (30, 6, "skewed", 4), (1500, 7, "skewed", 4),
                                                (400, 8, "drug", 1), (400, 9, "drug", 2), (400, 10, "drug", 3),
                                                (400, 11, "drug", 7), (5000, 12, "drug", 8), (60000, 13, "drug", 4)])
def test_kernel_built_index_arrays_and_plans_equal_the_torch_built_ones(dev, monkeypatch, n_mols, seed, dist, K):
    from mpnn_amd import synth
    mb = synth.make_molecules(n_mols, 8, seed=seed, dist=dist, edge_features=K)
    fast, slow = _pair(dev, mb, monkeypatch)
    _same(fast.edge_dst, slow.edge_dst, "edge_dst")
    _same(fast.order, slow.order, "order")
    _same(fast.type_ptr, slow.type_ptr, "type_ptr")
    _same(fast.transpose[0], slow.transpose[0], "t_row_ptr")
    _same(fast.transpose[1], slow.transpose[1], "t_eid")
    assert (fast.tile_plan is None) == (slow.tile_plan is None)
    if slow.tile_plan is not None:
        for name in ("tile_ptr", "tile_atom", "rt_ptr", "slots", "slot_eid", "tile_rec"):
            _same(getattr(fast.tile_plan, name), getattr(slow.tile_plan, name), "tile_plan." + name)
        assert fast.tile_plan.nbytes == slow.tile_plan.nbytes
    assert (fast.wide_plan is None) == (slow.wide_plan is None)
    if slow.wide_plan is not None:
        for name in ("tile_ptr", "tile_rec", "tile_atom", "blk_off", "slots", "slot_eid"):
            _same(getattr(fast.wide_plan, name), getattr(slow.wide_plan, name), "wide_plan." + name)
        assert fast.wide_plan.nbytes == slow.wide_plan.nbytes


def test_kernel_path_refuses_what_the_torch_path_refuses(dev, monkeypatch):
    """Two molecules of 100 atoms joined by one bond: inside one 256-atom index tile, across two 128-atom plan tiles -- the
    tile plan must refuse, the index arrays and the wide plan must not."""
    from mpnn_amd.graph import MolGraph
    n = 100
    src, dst = [], []
    for base in (0, n):
        for i in range(n - 1):
            src += [base + i, base + i + 1]
            dst += [base + i + 1, base + i]
    src += [n + 5, n - 1]
    dst += [n - 1, n + 5]
    s, d = np.array(src), np.array(dst)
    o = np.lexsort((s, d))
    s, d = s[o], d[o]
    row_ptr = np.zeros(2 * n + 1, np.int64)
    np.add.at(row_ptr, d + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    t = lambda a, dt: torch.from_numpy(np.asarray(a)).to(device=dev, dtype=dt)
    def graph():
        return MolGraph(t(row_ptr, torch.int32), t(s, torch.int32), None, torch.zeros(len(s), dtype=torch.int32, device=dev),
                        torch.eye(1, 4, device=dev), t([0, n, 2 * n], torch.int32))
    g = graph().prepare(tile_plan=True, wide_plan=True)
    assert g.tile_plan is None                               # an edge leaves its 128-atom tile
    from mpnn_amd import graph as G
    with monkeypatch.context() as m:
        m.setattr(G, "_fast_path", lambda *a, **k: False)
        ref = graph().prepare(tile_plan=True, wide_plan=True)
    assert ref.tile_plan is None
    assert torch.equal(g.transpose[1], ref.transpose[1]) and torch.equal(g.order, ref.order)
    assert (g.wide_plan is None) == (ref.wide_plan is None)
    # three such molecules in a row: the 256-atom index tiles are crossed too -> the index kernels hand back to the torch path
    big = synthetic_chain(dev, 3, n, cross=True)
    with monkeypatch.context() as m:
        m.setattr(G, "_fast_path", lambda *a, **k: False)
        ref2 = synthetic_chain(dev, 3, n, cross=True).prepare(tile_plan=False)
    big.prepare(tile_plan=False)
    assert torch.equal(big.transpose[1], ref2.transpose[1]) and torch.equal(big.edge_dst, ref2.edge_dst)


def synthetic_chain(dev, mols, n, cross):
    from mpnn_amd.graph import MolGraph
    src, dst = [], []
    for b in range(mols):
        base = b * n
        for i in range(n - 1):
            src += [base + i, base + i + 1]
            dst += [base + i + 1, base + i]
        if cross and b + 1 < mols:
            src += [base + n, base + n - 1]
            dst += [base + n - 1, base + n]
    s, d = np.array(src), np.array(dst)
    o = np.lexsort((s, d))
    s, d = s[o], d[o]
    row_ptr = np.zeros(mols * n + 1, np.int64)
    np.add.at(row_ptr, d + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    t = lambda a, dt: torch.from_numpy(np.asarray(a)).to(device=dev, dtype=dt)
    return MolGraph(t(row_ptr, torch.int32), t(s, torch.int32), None, torch.zeros(len(s), dtype=torch.int32, device=dev),
                    torch.eye(1, 4, device=dev), t(np.arange(0, mols * n + 1, n), torch.int32))
