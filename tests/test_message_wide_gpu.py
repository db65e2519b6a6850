"""The fused typed-message + neighbour-sum kernel at widths 128 / 256 (mpnn_message_aggregate_wide_f32,
csrc/message_tile_wide.hip: typed aggregate-then-contract on molecule tiles of up to 256 atoms) against float64: small
and ragged batches, configuration 5's hub molecules of up to 200 atoms, operand magnitudes far from 1 and rows of
very different magnitude inside one block (per-atom power-of-two scales, incl. the rescale of running accumulators),
bit reproducibility, the two-kernel path it replaces, and the autograd node that routes to it."""
import os

import numpy as np
import pytest
import torch

from conftest import max_err

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("MPNN_GRU_MATH") == "fp32" or bool(os.environ.get("MPNN_UNFUSED_MESSAGE")),
                                 reason="the fused kernels are switched off in this mode (ops.wide_kernel_applies)")]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _graph(dev, n_mols, F, seed, dist="drug", K=4):
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(n_mols, F, seed=seed, dist=dist, edge_features=K)
    return mb, MolGraph.from_molbatch(mb, dev), torch.from_numpy(mb.atom_feat).to(dev)


def _ref(g, h, A):
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    out = torch.zeros(g.num_nodes, A.shape[1], dtype=torch.float64, device=h.device)
    for k in range(A.shape[0]):                                 # one dense product per type: no (E, F, F) tensor
        idx = (typ == k).nonzero().squeeze(1)
        out.index_add_(0, dst[idx], h.double()[src[idx]] @ A.double()[k].t())
    return out


@pytest.mark.parametrize("F", [128, 256])
@pytest.mark.parametrize("n_mols,seed,dist", [(1, 1, "drug"), (3, 2, "drug"), (37, 3, "drug"), (3000, 4, "drug"),
                                              (1, 5, "skewed"), (40, 6, "skewed"), (1500, 7, "skewed")])
def test_wide_kernel_matches_float64(dev, F, n_mols, seed, dist):
    from mpnn_amd import ops
    mb, g, h = _graph(dev, n_mols, F, seed, dist)
    gen = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(g.num_types, F, F, device=dev, generator=gen) / F ** 0.5
    assert ops.wide_kernel_applies(A, None, None, g)
    assert not ops.wide_kernel_applies(A, None, torch.ones(g.num_edges, device=dev), g)      # weighted sums: two kernels
    out = ops.message_aggregate_wide_raw(h, A, g)
    ref = _ref(g, h, A)
    assert max_err(out, ref) < 1e-5 * max(1.0, float(ref.abs().max()))
    deg0 = (g.row_ptr[1:] == g.row_ptr[:-1])
    if bool(deg0.any()):
        assert float(out[deg0].abs().max()) == 0.0              # atoms without bonds: exact zeros


def test_wide_kernel_takes_hubs_of_200_atom_molecules(dev):
    """configs[4]: preferential-attachment molecules of up to 200 atoms, hubs with dozens of bonds, hidden 256."""
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 4000, 256, 317, "skewed")
    assert int(mb.n_atoms.max()) >= 190 and int((g.row_ptr[1:] - g.row_ptr[:-1]).max()) > 40
    gen = torch.Generator(device=dev).manual_seed(3)
    A = torch.randn(g.num_types, 256, 256, device=dev, generator=gen) / 16.0
    out = ops.message_aggregate_wide_raw(h, A, g)
    ref = _ref(g, h, A)
    assert max_err(out, ref) < 1e-5 * float(ref.abs().max())
    hub = int(torch.argmax(g.row_ptr[1:] - g.row_ptr[:-1]))
    assert max_err(out[hub], ref[hub]) < 1e-5 * float(ref[hub].abs().max())


@pytest.mark.parametrize("F", [128, 256])
@pytest.mark.parametrize("h_scale,a_scale", [(1e6, 1.0), (1e-6, 1.0), (1.0, 3e4), (1.0, 1e-5), (1e-20, 1e20), (3e8, 2e-9)])
def test_wide_kernel_is_scale_invariant(dev, F, h_scale, a_scale):
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 300, F, 11)
    gen = torch.Generator(device=dev).manual_seed(11)
    A = torch.randn(g.num_types, F, F, device=dev, generator=gen) / F ** 0.5
    out = ops.message_aggregate_wide_raw(h * h_scale, A * a_scale, g)
    ref = _ref(g, h * h_scale, A * a_scale)
    assert torch.isfinite(out).all()
    assert max_err(out, ref) / float(ref.abs().max()) < 2e-6


@pytest.mark.parametrize("F", [128, 256])
def test_wide_kernel_rows_and_columns_of_very_different_magnitude(dev, F):
    """Per-ATOM scales: atoms 1e6 apart inside one 32-atom block keep fp32-like accuracy per destination row; feature
    columns that grow 1000-fold along the contraction force the running scale down mid-tile (accumulators rescaled)."""
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 200, F, 12)
    gen = torch.Generator(device=dev).manual_seed(12)
    scale = torch.pow(10.0, torch.randint(-3, 4, (g.num_nodes, 1), device=dev, generator=gen).float())
    ramp = torch.pow(10.0, torch.linspace(-1.0, 2.5, F, device=dev)).unsqueeze(0)          # later chunks are larger
    hh = h * scale * ramp
    A = torch.randn(g.num_types, F, F, device=dev, generator=gen) / F ** 0.5
    A[1] *= 1e-3
    out = ops.message_aggregate_wide_raw(hh, A, g)
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    ref = _ref(g, hh, A)
    bound = torch.zeros(g.num_nodes, F, dtype=torch.float64, device=dev)
    for k in range(A.shape[0]):
        idx = (typ == k).nonzero().squeeze(1)
        bound.index_add_(0, dst[idx], hh.double()[src[idx]].abs() @ A.double()[k].abs().t())
    assert float(((out.double() - ref).abs() / (bound + 1e-300)).max()) < 4e-6


@pytest.mark.parametrize("F", [128, 256])
def test_wide_kernel_is_bit_reproducible_and_equals_the_two_kernel_path(dev, F):
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 8000, F, 13)
    gen = torch.Generator(device=dev).manual_seed(13)
    A = torch.randn(g.num_types, F, F, device=dev, generator=gen) / F ** 0.5
    a = ops.message_aggregate_wide_raw(h, A, g)
    b = ops.message_aggregate_wide_raw(h, A, g)
    assert torch.equal(a, b)
    two = ops.segsum_raw(ops.edge_message_raw(h, A, g), g.row_ptr, None, g.num_nodes)
    assert max_err(a, two) < 2e-5 * max(1.0, float(two.abs().max()))


@pytest.mark.parametrize("K", [1, 2, 4, 5, 7, 8])
@pytest.mark.parametrize("F,n_mols", [(128, 500), (128, 60000), (256, 12000)])
def test_bond_type_counts_small_and_large_batches(dev, K, F, n_mols):
    """Every type count the ops accept (graph.WidePlan.MAX_TYPES = 8), on a batch of a few tiles and on batches of more
    than 128 tiles (60k molecules = ~7.5k tiles, c4 / c5 sized): the plan's sort key used to wrap there at K = 7 and to
    overflow at K = 8."""
    from mpnn_amd import ops
    mb, g, h = _graph(dev, n_mols, F, 20 + K, K=K)
    assert g.num_types == K
    gen = torch.Generator(device=dev).manual_seed(K)
    A = torch.randn(K, F, F, device=dev, generator=gen) / F ** 0.5
    assert ops.wide_kernel_applies(A, None, None, g) and (n_mols < 10000 or g.wide_plan.num_tiles > 128)
    out = ops.message_aggregate_wide_raw(h, A, g)
    ref = _ref(g, h, A)
    assert max_err(out, ref) < 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("K", [4, 5, 8])
def test_gated_aggregate_falls_back_beyond_four_types(dev, K):
    """AttEdgeNetwork + AdjMsgAgg: the fused gated kernel covers K <= 4; beyond that ops.gated_message_aggregate takes the
    materialised gate + message_aggregate path (unit weights) and must still match float64."""
    from mpnn_amd import ops
    F = 128
    mb, g, h = _graph(dev, 400, F, 40 + K, K=K)
    gen = torch.Generator(device=dev).manual_seed(K)
    A = torch.randn(K, F, F, device=dev, generator=gen) / F ** 0.5
    z = torch.randn(g.num_nodes, F, device=dev, generator=gen)
    q = torch.randn(K, F, device=dev, generator=gen)
    assert ops.wide_gated_applies(A, None, g) == (K <= 4)
    out = ops.gated_message_aggregate(h, A, ops.LazyAttGate(z, q, g), g)
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    gate = torch.softmax(z.double()[dst] + q.double()[typ], dim=-1)
    x = gate * h.double()[src]
    ref = torch.zeros(g.num_nodes, F, dtype=torch.float64, device=dev)
    for k in range(K):
        idx = (typ == k).nonzero().squeeze(1)
        ref.index_add_(0, dst[idx], x[idx] @ A.double()[k].t())
    assert max_err(out, ref) < 1e-5 * max(1.0, float(ref.abs().max()))


def test_the_autograd_node(dev, monkeypatch):
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 2000, 128, 14)
    gen = torch.Generator(device=dev).manual_seed(14)
    A = (torch.randn(g.num_types, 128, 128, device=dev, generator=gen) / 11.0).requires_grad_(True)
    dagg = torch.randn(g.num_nodes, 128, device=dev, generator=gen)
    timer = ops.KernelTimer(["message_aggregate", "edge_message", "segsum"])
    ops.set_kernel_timer(timer)
    try:
        out = ops.message_aggregate(h, A, g)
        out.backward(dagg)
        assert len(timer.events["message_aggregate"]) == 1 and not timer.events["edge_message"]
        gA = A.grad.clone()
        monkeypatch.setenv("MPNN_UNFUSED_MESSAGE", "1")
        timer.reset()
        A.grad = None
        out2 = ops.message_aggregate(h, A, g)
        out2.backward(dagg)
        assert not timer.events["message_aggregate"] and len(timer.events["edge_message"]) == 1
    finally:
        ops.set_kernel_timer(None)
    assert max_err(out.detach(), out2.detach()) < 2e-5
    assert max_err(gA, A.grad) < 1e-5 * float(A.grad.abs().max())


# ------------------------------------------------------------------------------------------ width 64, resident matrices
@pytest.mark.parametrize("n_mols,seed,dist", [(1, 1, "drug"), (3, 2, "drug"), (37, 3, "drug"), (5000, 4, "drug"),
                                              (40, 6, "skewed"), (1500, 7, "skewed")])
def test_width64_resident_kernel_matches_float64(dev, n_mols, seed, dist):
    from mpnn_amd import ops
    mb, g, h = _graph(dev, n_mols, 64, seed, dist)
    gen = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(g.num_types, 64, 64, device=dev, generator=gen) / 8.0
    assert ops.wide_kernel_applies(A, None, None, g)
    out = ops.message_aggregate_wide_raw(h, A, g)
    ref = _ref(g, h, A)
    assert max_err(out, ref) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(out, ops.message_aggregate_wide_raw(h, A, g))           # bit-reproducible


@pytest.mark.parametrize("h_scale,a_scale", [(1e6, 1.0), (1e-6, 1.0), (1.0, 3e4), (1e-20, 1e20)])
def test_width64_resident_kernel_scales_and_rows_apart(dev, h_scale, a_scale):
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 300, 64, 11)
    gen = torch.Generator(device=dev).manual_seed(11)
    A = torch.randn(g.num_types, 64, 64, device=dev, generator=gen) / 8.0
    out = ops.message_aggregate_wide_raw(h * h_scale, A * a_scale, g)
    ref = _ref(g, h * h_scale, A * a_scale)
    assert torch.isfinite(out).all() and max_err(out, ref) / float(ref.abs().max()) < 2e-6
    scale = torch.pow(10.0, torch.randint(-3, 4, (g.num_nodes, 1), device=dev, generator=gen).float())
    hh = h * scale
    A2 = A.clone()
    A2[1] *= 1e-3
    out = ops.message_aggregate_wide_raw(hh, A2, g)
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    bound = torch.zeros(g.num_nodes, 64, dtype=torch.float64, device=dev)
    for k in range(A2.shape[0]):
        idx = (typ == k).nonzero().squeeze(1)
        bound.index_add_(0, dst[idx], hh.double()[src[idx]].abs() @ A2.double()[k].abs().t())
    assert float(((out.double() - _ref(g, hh, A2)).abs() / (bound + 1e-300)).max()) < 4e-6


def test_width64_batches_with_large_molecules_take_the_fused_path(dev):
    """A width-64 batch with molecules of more than 128 atoms does not fit the 128-atom tile kernel (message_tile.hip);
    the autograd node then takes the 256-atom resident-matrix kernel instead of the two-kernel path."""
    from mpnn_amd import ops
    mb, g, h = _graph(dev, 600, 64, 9, "skewed")
    assert int(mb.n_atoms.max()) > 128 and g.tile_plan is None and g.wide_plan is not None
    gen = torch.Generator(device=dev).manual_seed(9)
    A = torch.randn(g.num_types, 64, 64, device=dev, generator=gen) / 8.0
    timer = ops.KernelTimer(["message_aggregate", "edge_message", "segsum"])
    ops.set_kernel_timer(timer)
    try:
        out = ops.message_aggregate(h, A, g)
    finally:
        ops.set_kernel_timer(None)
    assert len(timer.events["message_aggregate"]) == 1 and not timer.events["edge_message"]
    ref = _ref(g, h, A)
    assert max_err(out, ref) < 1e-5 * max(1.0, float(ref.abs().max()))
