"""Kernel-level parity on a real MI355X, through the C ABI (ctypes) -- never through the oracle.

Integer / index work is checked bit-exact; fp32 work against the CPU oracle within 1e-5
(BASELINE.json north_star tolerance) on O(1) data.
"""
import numpy as np
import pytest
import torch

from conftest import Fixture, max_err
from oracle import dense_ref as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a device"
    from mpnn_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand_graph(rng, V, max_deg, skew=False):
    if skew:
        deg = np.minimum((rng.pareto(1.2, V) * 2).astype(np.int64), max_deg)
    else:
        deg = rng.integers(0, max_deg + 1, V)
    row_ptr = np.zeros(V + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    col = rng.integers(0, V, int(row_ptr[-1])).astype(np.int32)
    return row_ptr, col


# ----------------------------------------------------------------------------- dense -> CSR
def test_csr_matches_reference_fixture_bit_exact(dev):
    from mpnn_amd.graph import MolGraph
    f = Fixture("csr_ragged")
    adj = torch.from_numpy(f.raw["adj"]).to(dev)
    g = MolGraph.from_dense(adj, None)
    assert torch.equal(g.row_ptr.cpu(), torch.from_numpy(f.raw["row_ptr"]))
    assert torch.equal(g.col_idx.cpu(), torch.from_numpy(f.raw["col_idx"]))
    assert torch.equal(g.edge_weight.cpu(), torch.ones(g.num_edges))


@pytest.mark.parametrize("B,N,p", [(1, 1, 1.0), (3, 7, 0.0), (5, 64, 0.3), (4, 65, 0.5), (2, 200, 0.05), (700, 50, 0.04)])
def test_csr_random_bit_exact(dev, B, N, p):
    from mpnn_amd.graph import MolGraph
    g = torch.Generator().manual_seed(B * 1000 + N)
    adj = (torch.rand(B, N, N, generator=g) < p).float() * (torch.rand(B, N, N, generator=g) + 0.5)
    bfm = torch.zeros(B, N, N, 3)
    extra = torch.rand(B, N, N, generator=g) < p / 2          # pairs that only carry bond features
    bfm[extra] = torch.rand(int(extra.sum()), 3, generator=g) + 0.1
    for use_adj, use_bfm in ((True, False), (True, True), (False, True)):
        a = adj if use_adj else None
        b = bfm if use_bfm else None
        member = torch.zeros(B, N, N, dtype=torch.bool)
        if use_adj:
            member |= adj != 0
        if use_bfm:
            member |= (bfm != 0).any(-1)
        rp, ci, _ = O.dense_to_csr(member.float())
        mg = MolGraph.from_dense(a.to(dev) if a is not None else None, b.to(dev) if b is not None else None)
        assert torch.equal(mg.row_ptr.cpu(), rp)
        assert torch.equal(mg.col_idx.cpu(), ci)
        nz = member.nonzero()
        if use_adj:
            assert torch.equal(mg.edge_weight.cpu(), adj[nz[:, 0], nz[:, 1], nz[:, 2]])
        if use_bfm:
            assert torch.equal(mg.edge_feat.cpu(), bfm[nz[:, 0], nz[:, 1], nz[:, 2]])
            assert torch.equal(mg.type_feat[mg.edge_type.long()].cpu(), mg.edge_feat.cpu())


# ----------------------------------------------------------------------------- aggregator
def _segsum_oracle(msg, row_ptr, w):
    V = row_ptr.shape[0] - 1
    deg = np.diff(row_ptr)
    dst = torch.from_numpy(np.repeat(np.arange(V), deg))
    m = msg.double() * (w.double().unsqueeze(-1) if w is not None else 1.0)
    return torch.zeros(V, msg.shape[1], dtype=torch.float64).index_add_(0, dst, m)


@pytest.mark.parametrize("F", [1, 3, 8, 22, 64, 100, 128, 256, 260, 512])
@pytest.mark.parametrize("weighted", [False, True])
def test_segsum(dev, F, weighted):
    from mpnn_amd import ops
    rng = np.random.default_rng(F)
    row_ptr, _ = _rand_graph(rng, 777, 9, skew=(F % 2 == 0))
    E = int(row_ptr[-1])
    msg = torch.from_numpy(rng.standard_normal((E, F)).astype(np.float32))
    w = torch.from_numpy(rng.random(E).astype(np.float32)) if weighted else None
    out = ops.segsum_raw(msg.to(dev), torch.from_numpy(row_ptr).to(dev), w.to(dev) if weighted else None, 777)
    ref = _segsum_oracle(msg, row_ptr, w)
    assert max_err(out.cpu(), ref) < TOL
    # rows without edges are exactly zero
    empty = np.diff(row_ptr) == 0
    assert float(out.cpu()[torch.from_numpy(empty)].abs().sum()) == 0.0
    # backward kernel: dmsg[e] = w[e] * dout[dst(e)]
    dout = torch.from_numpy(rng.standard_normal((777, F)).astype(np.float32))
    dmsg = ops.segsum_bwd_raw(dout.to(dev), torch.from_numpy(row_ptr).to(dev), w.to(dev) if weighted else None, E)
    dst = np.repeat(np.arange(777), np.diff(row_ptr))
    exp = dout[dst] * (w.unsqueeze(-1) if weighted else 1.0)
    assert max_err(dmsg.cpu(), exp) == 0.0


def test_segsum_empty_and_long_rows(dev):
    from mpnn_amd import ops
    rng = np.random.default_rng(9)
    deg = np.array([0, 0, 1500, 0, 3, 1, 0, 0], np.int64)       # one hub, ragged tail
    row_ptr = np.zeros(9, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    msg = torch.from_numpy(rng.standard_normal((int(row_ptr[-1]), 64)).astype(np.float32))
    out = ops.segsum_raw(msg.to(dev), torch.from_numpy(row_ptr).to(dev), None, 8)
    ref = _segsum_oracle(msg, row_ptr, None)
    assert max_err(out.cpu(), ref) < TOL * float(ref.abs().max())        # 1500-term fp32 sums, |sum| ~ 100
    z = ops.segsum_raw(torch.zeros(0, 64, device=dev), torch.zeros(5, dtype=torch.int32, device=dev), None, 4)
    assert z.shape == (4, 64) and float(z.abs().sum()) == 0.0


def test_segsum_gather(dev):
    from mpnn_amd import ops
    rng = np.random.default_rng(10)
    row_ptr, col = _rand_graph(rng, 500, 6)
    x = torch.from_numpy(rng.standard_normal((500, 64)).astype(np.float32))
    out = ops.segsum_gather_raw(x.to(dev), torch.from_numpy(row_ptr).to(dev), torch.from_numpy(col).to(dev), None, 500)
    assert max_err(out.cpu(), _segsum_oracle(x[col.astype(np.int64)], row_ptr, None)) < TOL


# ----------------------------------------------------------------------------- edge message
def _message_setup(rng, V, K, nf, mf, max_deg=5):
    from mpnn_amd.graph import MolGraph
    row_ptr, col = _rand_graph(rng, V, max_deg)
    E = int(row_ptr[-1])
    et = rng.integers(0, K, E).astype(np.int32)
    if K > 2:
        et[et == 1] = 0                 # leave one type empty
    h = rng.standard_normal((V, nf)).astype(np.float32)
    A = (rng.standard_normal((K, mf, nf)) / np.sqrt(nf)).astype(np.float32)
    return row_ptr, col, et, h, A


@pytest.mark.parametrize("nf,mf,K,V", [(8, 8, 4, 50), (22, 22, 5, 333), (64, 64, 4, 2000), (64, 32, 3, 700),
                                       (100, 72, 2, 300), (128, 128, 4, 900), (256, 256, 3, 400), (64, 64, 1, 129),
                                       (64, 64, 70, 900), (22, 22, 600, 500),       # 64 < K <= 4096: staged kernel
                                       (64, 64, 5000, 2600), (24, 40, 4500, 2400), (130, 70, 4200, 2300)])   # per-type matvec
@pytest.mark.parametrize("gated", [False, True])
def test_edge_message(dev, nf, mf, K, V, gated):
    from mpnn_amd import ops
    from mpnn_amd.graph import MolGraph
    rng = np.random.default_rng(nf * 7 + mf + K)
    row_ptr, col, et, h, A = _message_setup(rng, V, K, nf, mf)
    E = int(row_ptr[-1])
    t = lambda a, dt=None: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(col), None, t(et), torch.zeros(K, 1, device=dev),
                 torch.tensor([0, V], dtype=torch.int32, device=dev))
    gate = rng.random((E, nf)).astype(np.float32) if gated else None
    msg = ops.edge_message_raw(t(h), t(A), g, t(gate) if gated else None)
    x = torch.from_numpy(h)[col.astype(np.int64)].double()
    if gated:
        x = x * torch.from_numpy(gate).double()
    ref = torch.einsum("emn,en->em", torch.from_numpy(A)[et.astype(np.int64)].double(), x)
    assert msg.shape == (E, mf)
    assert max_err(msg.cpu(), ref) < TOL * max(1.0, float(ref.abs().max()))
    # type-sorted order really is a permutation grouped by type
    order = g.order.cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(E))
    assert (np.diff(et[order]) >= 0).all()


def test_edge_message_integer_exact(dev):
    """Small-integer data: fp32 products and sums are exact, so any indexing slip shows as != 0."""
    from mpnn_amd import ops
    from mpnn_amd.graph import MolGraph
    rng = np.random.default_rng(77)
    V, K, nf, mf = 600, 4, 64, 64
    row_ptr, col, et, _, _ = _message_setup(rng, V, K, nf, mf)
    h = rng.integers(-3, 4, (V, nf)).astype(np.float32)
    A = rng.integers(-2, 3, (K, mf, nf)).astype(np.float32)          # asymmetric by construction
    t = lambda a: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(col), None, t(et), torch.zeros(K, 1, device=dev),
                 torch.tensor([0, V], dtype=torch.int32, device=dev))
    msg = ops.edge_message_raw(t(h), t(A), g)
    ref = torch.einsum("emn,en->em", torch.from_numpy(A)[et.astype(np.int64)], torch.from_numpy(h)[col.astype(np.int64)])
    assert torch.equal(msg.cpu(), ref)


# ----------------------------------------------------------------------------- GRU
@pytest.mark.parametrize("tag", ["h8", "h22", "h64"])
def test_gru_against_reference_fixture(dev, tag):
    from mpnn_amd import ops
    f = Fixture("gru_update_" + tag)
    H = f.inputs["node_states"].shape[-1]
    p = {k: v.to(dev) for k, v in f.params.items()}
    out, saved = ops.gru_update_raw(f.inputs["messages"].reshape(-1, H).to(dev),
                                    f.inputs["node_states"].reshape(-1, H).contiguous().to(dev),
                                    f.inputs["mask"].reshape(-1).to(dev), p["gru_cell.weight_ih"],
                                    p["gru_cell.weight_hh"], p["gru_cell.bias_ih"], p["gru_cell.bias_hh"], True)
    assert max_err(out.cpu().view(f.out[""].shape), f.out[""]) < TOL
    pad = f.inputs["mask"].reshape(-1) == 0
    assert float(out.cpu()[pad].abs().sum()) == 0.0          # padded atoms exactly zero
    assert saved.shape == (out.shape[0], 4 * H)


@pytest.mark.parametrize("V,H", [(1, 64), (127, 64), (129, 64), (1000, 128), (300, 256), (257, 100), (513, 48)])
def test_gru_random(dev, V, H):
    from mpnn_amd import ops
    g = torch.Generator().manual_seed(V + H)
    m = torch.rand(V, H, generator=g) * 2 - 1
    h = torch.rand(V, H, generator=g) * 2 - 1
    mask = (torch.rand(V, generator=g) < 0.8).float()
    s = 1.0 / np.sqrt(H)
    p = {"gru_cell.weight_ih": (torch.rand(H, 3 * H, generator=g) * 2 - 1) * s,
         "gru_cell.weight_hh": (torch.rand(H, 3 * H, generator=g) * 2 - 1) * s,
         "gru_cell.bias_ih": torch.rand(3 * H, generator=g) - 0.5,
         "gru_cell.bias_hh": torch.rand(3 * H, generator=g) - 0.5}
    ref = O.gru_update(p, m, h, mask.view(-1, 1))
    out, _ = ops.gru_update_raw(m.to(dev), h.to(dev), mask.to(dev), *(p[k].to(dev) for k in
                                ("gru_cell.weight_ih", "gru_cell.weight_hh", "gru_cell.bias_ih", "gru_cell.bias_hh")),
                                False)
    assert max_err(out.cpu(), ref) < TOL


# ----------------------------------------------------------------------------- edge tower chain
@pytest.mark.parametrize("R,L,n", [(5, 256, 50), (1, 16, 50), (9, 49, 50), (21, 200, 7), (64, 256, 3)])
def test_tower_chain(dev, R, L, n):
    """The 50 aliased Linear(L,L,bias=False)+ReLU layers in one kernel vs the layer-by-layer product,
    forward and both gradients (float64 reference)."""
    from mpnn_amd import ops
    g = torch.Generator().manual_seed(R * 1000 + L + n)
    W = (torch.rand(L, L, generator=g) * 2 - 1) * (1.9 / L) ** 0.5 * 1.7      # keeps activations O(1) over 50 layers
    x = torch.rand(R, L, generator=g)
    cot = torch.rand(R, L, generator=g) - 0.5
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    y = xd
    for _ in range(n):
        y = torch.relu(y @ Wd.t())
    gx, gW = torch.autograd.grad((y * cot.double()).sum(), [xd, Wd])
    xg, Wg = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    out = ops.tower_chain(xg, Wg, n)
    hx, hW = torch.autograd.grad((out * cot.to(dev)).sum(), [xg, Wg])
    scale = max(1.0, float(y.detach().abs().max()))
    assert max_err(out.cpu(), y) < 2e-5 * scale
    assert max_err(hx.cpu(), gx) < 2e-5 * max(1.0, float(gx.abs().max()))
    assert max_err(hW.cpu(), gW) < 2e-5 * max(1.0, float(gW.abs().max()))


# ----------------------------------------------------------------------------- masked batch norm
def test_masked_bn_against_reference_fixtures(dev):
    from mpnn_amd import ops
    f = Fixture("mask_bn1d_train")
    x, mask = f.inputs["x"].reshape(-1, 8).to(dev), f.inputs["mask"].reshape(-1).to(dev)
    y, mean, var = ops.masked_batch_norm(x, mask, f.params["weight"].to(dev), f.params["bias"].to(dev), None, 1e-5,
                                         ops.BN_MASKED_MEAN)
    assert max_err(y.cpu().view(f.out[""].shape), f.out[""]) < TOL
    # running statistics after one training step from (0, 1) with momentum 0.1
    assert max_err(0.1 * mean.cpu(), torch.from_numpy(f.raw["running_mean_after"])) < TOL
    assert max_err(0.9 + 0.1 * var.cpu(), torch.from_numpy(f.raw["running_var_after"])) < TOL
    f = Fixture("mask_bn1d_eval")
    y, _, _ = ops.masked_batch_norm(x, mask, f.params["weight"].to(dev), f.params["bias"].to(dev),
                                    (f.params["running_mean"].to(dev), f.params["running_var"].to(dev)), 1e-5,
                                    ops.BN_MASKED_MEAN | ops.BN_USE_STATS)
    assert max_err(y.cpu().view(f.out[""].shape), f.out[""]) < TOL
    f = Fixture("mask_bn_noaffine")
    y, _, _ = ops.masked_batch_norm(x, mask, None, None, None, 1e-6, ops.BN_EPS_INSIDE)
    assert max_err(y.cpu().view(f.out[""].shape), f.out[""]) < TOL


@pytest.mark.parametrize("V,F", [(37, 8), (1000, 22), (5000, 64), (777, 128), (300, 260)])
@pytest.mark.parametrize("variant", ["bn1d", "noaffine"])
def test_masked_bn_forward_backward(dev, V, F, variant):
    """Forward and all gradients against the oracle's differentiable restatement (float64)."""
    from mpnn_amd import ops
    g = torch.Generator().manual_seed(V + F)
    x = torch.randn(V, F, generator=g) * 2 + 0.5
    mask = (torch.rand(V, generator=g) < 0.7).float()
    x = x * (mask.unsqueeze(1) if variant == "noaffine" else 1.0) + (0.0 if variant == "noaffine" else 0.0)
    w, b = torch.rand(F, generator=g) + 0.5, torch.rand(F, generator=g) - 0.5
    cot = torch.randn(V, F, generator=g)
    xd = x.double().requires_grad_(True)
    if variant == "bn1d":
        wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
        ref, _, _ = O.mask_bn1d(xd, mask.double().view(-1, 1), wd, bd, None, None, True)
        gref = torch.autograd.grad((ref * cot.double()).sum(), [xd, wd, bd])
    else:
        ref = O.mask_bn(xd, mask.double().view(-1, 1))
        gref = torch.autograd.grad((ref * cot.double()).sum(), [xd])
    xg = x.to(dev).requires_grad_(True)
    if variant == "bn1d":
        wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        out, _, _ = ops.masked_batch_norm(xg, mask.to(dev), wg, bg, None, 1e-5, ops.BN_MASKED_MEAN)
        ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), [xg, wg, bg])
    else:
        out, _, _ = ops.masked_batch_norm(xg, mask.to(dev), None, None, None, 1e-6, ops.BN_EPS_INSIDE)
        ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), [xg])
    assert max_err(out.cpu(), ref) < 2e-5
    for a, bb in zip(ggpu, gref):
        assert max_err(a.cpu(), bb) < 2e-5 * max(1.0, float(bb.abs().max()))


# ----------------------------------------------------------------------------- attention gate
@pytest.mark.parametrize("V,F,K", [(300, 128, 4), (1000, 64, 5), (77, 8, 3), (500, 24, 40), (400, 256, 2), (900, 100, 700)])
def test_att_gate_forward_backward(dev, V, F, K):
    """gate[e] = softmax_f(z_atom[dst e] + q[type e]) and its backward (dz_atom by destination row, dq by type)
    against float64 autograd of the same expression."""
    from mpnn_amd import ops
    from mpnn_amd.graph import MolGraph
    rng = np.random.default_rng(V + F + K)
    row_ptr, col = _rand_graph(rng, V, 6)
    E = int(row_ptr[-1])
    et = rng.integers(0, K, E).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(col), None, t(et), torch.zeros(K, 1, device=dev),
                 torch.tensor([0, V], dtype=torch.int32, device=dev))
    z = torch.from_numpy(rng.standard_normal((V, F)).astype(np.float32) * 2).to(dev).requires_grad_(True)
    q = torch.from_numpy(rng.standard_normal((K, F)).astype(np.float32)).to(dev).requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((E, F)).astype(np.float32)).to(dev)
    gate = ops.att_gate(z, q, g)
    gate.backward(cot)
    z64, q64 = z.detach().double().requires_grad_(True), q.detach().double().requires_grad_(True)
    ref = torch.softmax(z64[g.edge_dst.long()] + q64[g.edge_type.long()], dim=-1)
    ref.backward(cot.double())
    assert max_err(gate.detach(), ref.detach()) < 2e-6
    assert max_err(z.grad, z64.grad) < 1e-5
    assert max_err(q.grad, q64.grad) / max(1.0, float(q64.grad.abs().max())) < 1e-5


@pytest.mark.parametrize("H", [64, 128, 256])
@pytest.mark.parametrize("profile", ["unit", "rows_1e-6_to_10", "growing_along_k", "h_much_larger", "tiny", "zero_rows"])
def test_gru_forward_wide_row_guards(dev, H, profile):
    """The GRU forward runs on two fp16 pieces per operand; each atom's m | h row is range-guarded by its own power-of-two
    scale.  Widths 128 / 256 (gru_update_stream_wide_kernel<.., F16>): chosen at the first K chunk with eight-fold headroom
    and lowered (the row's accumulator entries rescaled) when a later chunk outgrows it.  Width 64
    (gru_update_split_kernel<.., F16>): the whole row is in registers, m and h rows get separate scales and the r / z
    accumulators are rescaled between the two products.  Rows of wildly
    different magnitude in one tile, entries that grow along K, h rows far above m rows, tiny and all-zero rows must keep
    the float32 bar on the gate PRE-activations' scale, i.e. on `out` and on the saved gates."""
    from mpnn_amd import ops
    V = 20_011
    g = torch.Generator(device=dev).manual_seed(H + len(profile))
    m = torch.randn(V, H, device=dev, generator=g)
    h = torch.rand(V, H, device=dev, generator=g) * 2 - 1
    mask = (torch.rand(V, device=dev, generator=g) > 0.1).float()
    bound = (6.0 / (4 * H)) ** 0.5
    W_ih, W_hh = ((torch.rand(H, 3 * H, device=dev, generator=g) * 2 - 1) * bound for _ in range(2))
    b_ih, b_hh = (torch.rand(3 * H, device=dev, generator=g) * 0.2 - 0.1 for _ in range(2))
    if profile == "rows_1e-6_to_10":                          # neighbouring rows 7 orders of magnitude apart (larger rows
        f = torch.logspace(-6, 1, 8, device=dev)[torch.arange(V, device=dev) % 8].reshape(-1, 1)    # would only test
        m = m * f                                             # float32 conditioning: pre-activations of 1e3)
    elif profile == "growing_along_k":                        # later K chunks outgrow the first one 100-fold
        m = m * torch.logspace(-2, 0.5, H, device=dev).reshape(1, -1)
    elif profile == "h_much_larger":
        m, h = m * 1e-3, h * 30.0
        W_hh = W_hh * 0.03
    elif profile == "tiny":
        m, h = m * 1e-12, h * 1e-12
    elif profile == "zero_rows":
        m[::3] = 0.0
        h[::3] = 0.0
    m, h = m.contiguous(), h.contiguous()
    out, saved = ops.gru_update_raw(m, h, mask, W_ih, W_hh, b_ih, b_hh, True)
    gi = m.double() @ W_ih.double() + b_ih.double()
    gh = h.double() @ W_hh.double() + b_hh.double()
    mk = mask.double().reshape(-1, 1)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mk
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mk
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mk
    ref = ((1 - z) * n + z * h.double()) * mk
    scale = max(1.0, float(h.abs().max()))
    assert float((out.double() - ref).abs().max()) < 1e-5 * scale, profile
    sv = saved.double().reshape(V, 4, H)
    assert float((sv[:, 0] - r).abs().max()) < 1e-5
    assert float((sv[:, 1] - z).abs().max()) < 1e-5
    assert float((sv[:, 2] - n).abs().max()) < 1e-5
    nh_ref = gh[:, 2 * H:]
    assert float((sv[:, 3] - nh_ref).abs().max()) < 1e-5 * max(1.0, float(nh_ref.abs().max()))
    assert float(out[mask == 0].abs().max()) == 0.0
