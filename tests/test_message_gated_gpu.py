"""AttEdgeNetwork followed by the sum aggregator as ONE kernel, the feature gate formed inside it
(mpnn_message_aggregate_wide_gated_f32; reference: mpnn_functions/message/att_edge_network.py:18-31 composed with
message_aggregators/adjacent_message_agg.py:18).  The gate softmax_f(W_h h_i + W_e e_ij + b) depends on the destination
atom and the bond type only, so out[i] = sum_k A_k (g_ik * S_k[i]).  Checked against float64 and against the unfused
kernels (gate tensor, gated message rows, segmented sum)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    if os.environ.get("MPNN_GRU_MATH") == "fp32":
        pytest.skip("the fused message + sum kernels are split-precision kernels")
    return torch.device("cuda:0")


def _batch(dev, mols, dist="drug", seed=3):
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(mols, 128, seed=seed, dist=dist)
    return mb, MolGraph.from_molbatch(mb, dev)


def _ref64(h, A, z, q, g):
    """float64: per edge gate = softmax(z[dst] + q[type]); out[dst] += A[type] (gate * h[src])."""
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    gate = torch.softmax(z.double()[dst] + q.double()[typ], dim=-1)
    x = gate * h.double()[src]
    msg = torch.einsum("emn,en->em", A.double()[typ], x)
    return torch.zeros(h.shape[0], A.shape[1], dtype=torch.float64, device=h.device).index_add(0, dst, msg)


@pytest.mark.parametrize("mols,dist", [(1, "drug"), (37, "drug"), (700, "drug"), (60, "skewed")])
@pytest.mark.parametrize("zscale", [1.0, 30.0])
def test_gated_fused_forward_and_gradients(dev, mols, dist, zscale):
    from mpnn_amd import ops
    mb, g = _batch(dev, mols, dist)
    V, K, F = g.num_nodes, g.num_types, 128
    gen = torch.Generator(device=dev).manual_seed(5 + mols)
    h = torch.randn(V, F, device=dev, generator=gen)
    A = (torch.randn(K, F, F, device=dev, generator=gen) / F ** 0.5).requires_grad_(True)
    z = (torch.randn(V, F, device=dev, generator=gen) * zscale).requires_grad_(True)     # zscale 30: near one-hot gates
    q = (torch.randn(K, F, device=dev, generator=gen) * zscale).requires_grad_(True)
    cot = torch.randn(V, F, device=dev, generator=gen)
    assert ops.wide_gated_applies(A, None, g)
    out = ops.gated_message_aggregate(h, A, ops.LazyAttGate(z, q, g), g)
    (out * cot).sum().backward()
    got = (out.detach(), A.grad.clone(), z.grad.clone(), q.grad.clone())

    A64, z64, q64 = (t.detach().double().requires_grad_(True) for t in (A, z, q))
    ref = _ref64(h, A64, z64, q64, g)
    (ref * cot.double()).sum().backward()
    want = (ref.detach(), A64.grad, z64.grad, q64.grad)
    for name, a, b in zip(("out", "dA", "dz_atom", "dq"), got, want):
        err = float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
        assert err < (2e-5 if name == "out" else 1e-4), (name, err)

    # the unfused kernels on the same inputs
    A2, z2, q2 = (t.detach().clone().requires_grad_(True) for t in (A, z, q))
    o2 = ops.message_aggregate(h, A2, g, None, ops.att_gate(z2, q2, g))
    (o2 * cot).sum().backward()
    assert float((out.detach() - o2.detach()).abs().max() / o2.detach().abs().max()) < 2e-5
    for a, b in ((A.grad, A2.grad), (z.grad, z2.grad), (q.grad, q2.grad)):
        assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < 1e-4   # (near one-hot gates: u - g D cancels)


@pytest.mark.parametrize("K", [1, 2, 3])
@pytest.mark.parametrize("spread", [0.0, 6.0])
def test_gated_backward_per_atom_and_type(dev, K, spread):
    """The per-(atom, type) backward (mpnn_message_aggregate_wide_gated_bwd_f32 + mpnn_edge_message_agg_bwd_da_att_f32)
    against float64 and against the per-edge kernels, at every bond-type count below four, several tiles, and with the rows
    of the incoming gradient spread over 10^+-spread (the kernel splits every row behind its own power-of-two scale)."""
    from mpnn_amd import ops, synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(900, 128, seed=40 + K, edge_features=K)
    g = MolGraph.from_molbatch(mb, dev)
    V, F = g.num_nodes, 128
    assert g.num_types == K
    gen = torch.Generator(device=dev).manual_seed(17 + K)
    h = torch.randn(V, F, device=dev, generator=gen)
    A0 = torch.randn(K, F, F, device=dev, generator=gen) / F ** 0.5
    z0 = torch.randn(V, F, device=dev, generator=gen) * 3
    q0 = torch.randn(K, F, device=dev, generator=gen) * 3
    cot = torch.randn(V, F, device=dev, generator=gen)
    if spread:
        cot = cot * torch.pow(10.0, (torch.rand(V, 1, device=dev, generator=gen) * 2 - 1) * spread)
    res = []
    for per_edge in (False, True):
        A, z, q = (t.clone().requires_grad_(True) for t in (A0, z0, q0))
        ops.ATT_BWD_PER_EDGE = per_edge
        try:
            out = ops.gated_message_aggregate(h, A, ops.LazyAttGate(z, q, g), g)
            (out * cot).sum().backward()
        finally:
            ops.ATT_BWD_PER_EDGE = False
        res.append((A.grad, z.grad, q.grad))
    A64, z64, q64 = (t.double().requires_grad_(True) for t in (A0, z0, q0))
    (_ref64(h, A64, z64, q64, g) * cot.double()).sum().backward()
    want = (A64.grad, z64.grad, q64.grad)
    for name, new, old, ref in zip(("dA", "dz_atom", "dq"), res[0], res[1], want):
        scale = ref.abs().max().clamp_min(1e-30)
        e_new, e_old = float((new.double() - ref).abs().max() / scale), float((old.double() - ref).abs().max() / scale)
        assert e_new < 2e-5, (name, e_new, e_old)
    # per ROW of dz_atom: an atom whose incoming gradient is 10^-spread of the largest keeps its own accuracy.  dz_atom[i] is
    # linear in cot[i], so the yardstick of row i is |cot[i]| times the largest gain any row shows (the rows' own maxima are
    # no yardstick: u - g D cancels to almost nothing where a gate is nearly one-hot)
    cot_row = cot.double().abs().amax(1).clamp_min(1e-300)
    gain = (want[1].abs().amax(1) / cot_row).max()
    rel = (res[0][1].double() - want[1]).abs().amax(1) / (cot_row * gain)
    assert float(rel.max()) < 2e-5, float(rel.max())


def test_attention_model_takes_the_gated_kernel(dev):
    """models/att_model.py at hidden 128: the message functions hand the aggregator a lazy gate; outputs and gradients with
    the fused gated kernel equal those of the two-kernel path (MPNN_UNFUSED_MESSAGE=1)."""
    from mpnn_amd import ops, synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.att_model import BasicModel as AttModel
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    H, T = 128, 3
    mb = synth.make_molecules(200, H, seed=21)
    torch.manual_seed(8)
    model = AttModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T,
                     readout_func=GraphLevelOutput).to(dev)
    g = MolGraph.from_molbatch(mb, dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    mask = torch.ones(afm.shape[0], 1, device=dev)
    calls = []
    real = ops.message_aggregate_wide_gated_raw
    ops.message_aggregate_wide_gated_raw = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    res = []
    try:
        for unfused in (False, True):
            if unfused:
                os.environ["MPNN_UNFUSED_MESSAGE"] = "1"
            model.zero_grad()
            out = model(afm, g, g, mask)
            out.square().sum().backward()
            res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        os.environ.pop("MPNN_UNFUSED_MESSAGE", None)
        ops.message_aggregate_wide_gated_raw = real
    assert len(calls) == T                                            # one fused gated launch per step, fused run only
    assert float((res[0][0] - res[1][0]).abs().max() / res[1][0].abs().max()) < 5e-5
    for k, gr in res[0][1].items():
        ref = res[1][1][k]
        assert float((gr - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) < 5e-4, k


def test_attention_model_without_bonds_and_on_a_dense_batch(dev):
    """Edge cases of the lazy gate: molecules of one atom (no edge at all: the fused kernel has nothing to launch, the
    backward takes the generic route) and a dense padded batch (no tile plan: the gate is materialised as before)."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.att_model import BasicModel as AttModel
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    import numpy as np
    H = 128
    torch.manual_seed(3)
    model = AttModel(H, 4, H, 6, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=2,
                     readout_func=GraphLevelOutput).to(dev)
    # (a) 40 single-atom molecules, compact batch
    mb = synth.make_molecules(40, H, seed=2)
    singles = synth.select(mb, np.arange(40))
    one = synth.MolBatch(n_atoms=np.ones(40, np.int32), atom_ptr=np.arange(41, dtype=np.int32),
                         row_ptr=np.zeros(41, np.int32), col_idx=np.zeros(0, np.int32),
                         bond_type=np.zeros(0, singles.bond_type.dtype), type_feat=singles.type_feat,
                         edge_feat=None if singles.edge_feat is None else singles.edge_feat[:0],
                         atom_feat=singles.atom_feat[:40].copy())
    g = MolGraph.from_molbatch(one, dev)
    afm = torch.from_numpy(one.atom_feat).to(dev)
    out = model(afm, g, g, torch.ones(40, 1, device=dev))
    out.sum().backward()
    assert torch.isfinite(out).all() and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # (b) dense padded batch of the same width
    d = synth.to_dense(synth.select(mb, np.arange(6)))
    dense = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k in ("afm", "bfm", "adj", "mask")}
    model.zero_grad()
    out = model(dense["afm"], dense["bfm"], dense["adj"], dense["mask"])
    out.sum().backward()
    assert torch.isfinite(out).all()


def test_bias_gradient_of_the_logits_linear_takes_the_known_column_sums(dev):
    """sum_i dz_atom[i] = sum_k dq[k]: the Linear behind z_atom gets its bias gradient from the gated backward's dq instead
    of reducing (V, F) again (ops._offer_colsum / _take_colsum) -- same values as the reduction, and the offer is refused
    for any other tensor."""
    from mpnn_amd import ops
    mb, g = _batch(dev, 300)
    V, K, F = g.num_nodes, g.num_types, 128
    gen = torch.Generator(device=dev).manual_seed(9)
    h = torch.randn(V, F, device=dev, generator=gen)
    A = (torch.randn(K, F, F, device=dev, generator=gen) / F ** 0.5).requires_grad_(True)
    W = (torch.randn(F, F, device=dev, generator=gen) / F ** 0.5).requires_grad_(True)
    q = torch.randn(K, F, device=dev, generator=gen).requires_grad_(True)
    cot = torch.randn(V, F, device=dev, generator=gen)
    taken, real = [], ops._take_colsum
    res = []
    for use in (True, False):
        b = torch.zeros(F, device=dev, requires_grad=True)
        ops._take_colsum = (lambda t: (taken.append(real(t)), taken[-1])[1]) if use else (lambda t: None)
        try:
            z = ops.tall_linear(h, W, b)
            out = ops.gated_message_aggregate(h, A, ops.LazyAttGate(z, q, g), g)
            (out * cot).sum().backward()
        finally:
            ops._take_colsum = real
        res.append(b.grad.clone())
    assert len(taken) == 1 and taken[0] is not None            # the offer was found and accepted
    assert float((res[0] - res[1]).abs().max() / res[1].abs().max()) < 2e-5
    other = torch.zeros(V, F, device=dev)
    ops._offer_colsum(other, torch.ones(F, device=dev))
    other.add_(1.0)                                             # modified since the offer: refused
    assert ops._take_colsum(other) is None
