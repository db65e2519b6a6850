"""Gradients of the HIP path against gradients recorded from the REAL reference (g.in.* / g.p.* in
tests/golden/*.npz: d(sum(out * cot)) / d(input | parameter))."""
import numpy as np
import pytest
import torch

from conftest import Fixture, max_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    return max_err(a, b) / max(1.0, float(torch.as_tensor(b).abs().max()))


def _check_param_grads(module, f, tol):
    seen = 0
    for k, p in module.named_parameters():
        if k in f.gp:
            assert p.grad is not None, k
            assert _rel(p.grad.cpu(), f.gp[k]) < tol, (k, _rel(p.grad.cpu(), f.gp[k]))
            seen += 1
    assert seen > 0


@pytest.mark.parametrize("tag,H", [("h8", 8), ("h22", 22), ("h64", 64)])
def test_gru_update_backward(dev, tag, H):
    from mpnn_amd.mpnn_functions import GRUUpdate
    f = Fixture("gru_update_" + tag)
    m = GRUUpdate(H, H).to(dev)
    m.load_state_dict(f.params)
    msg = f.inputs["messages"].to(dev).requires_grad_(True)
    h = f.inputs["node_states"].to(dev).requires_grad_(True)
    out = m(msg, h, f.inputs["mask"].to(dev))
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(msg.grad.cpu(), f.gin["messages"]) < TOL
    assert _rel(h.grad.cpu(), f.gin["node_states"]) < TOL
    _check_param_grads(m, f, TOL)


@pytest.mark.parametrize("V,H", [(300, 64), (1000, 128), (130, 256), (77, 40)])
def test_gru_backward_random(dev, V, H):
    from oracle import dense_ref as O
    from mpnn_amd import ops
    g = torch.Generator().manual_seed(V * 3 + H)
    mk = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)
    m, h, cot = mk(V, H), mk(V, H), mk(V, H)
    mask = (torch.rand(V, generator=g) < 0.8).float()
    s = 1.0 / np.sqrt(H)
    names = ("gru_cell.weight_ih", "gru_cell.weight_hh", "gru_cell.bias_ih", "gru_cell.bias_hh")
    p = {names[0]: mk(H, 3 * H) * s, names[1]: mk(H, 3 * H) * s, names[2]: mk(3 * H) * 0.5, names[3]: mk(3 * H) * 0.5}
    leaves = [t.clone().requires_grad_(True) for t in (m, h, *[p[n] for n in names])]
    ref = O.gru_update(dict(zip(names, leaves[2:])), leaves[0], leaves[1], mask.view(-1, 1))
    gref = torch.autograd.grad((ref * cot).sum(), leaves)
    dl = [t.clone().to(dev).requires_grad_(True) for t in (m, h, *[p[n] for n in names])]
    out = ops.gru_update(dl[0], dl[1], mask.to(dev), *dl[2:])
    ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), dl)
    for a, b, name in zip(ggpu, gref, ("dm", "dh") + names):
        assert _rel(a.cpu(), b) < 2e-5, (name, _rel(a.cpu(), b))


def _gru_bwd_ref64(m, h, mask, dout, W1, W2, b1, b2):
    H = m.shape[1]
    m64, h64 = m.double().requires_grad_(True), h.double().requires_grad_(True)
    W1d, W2d, b1d, b2d = (x.double().requires_grad_(True) for x in (W1, W2, b1, b2))
    gi, gh = m64 @ W1d + b1d, h64 @ W2d + b2d
    mk = mask.double().reshape(-1, 1)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mk
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mk
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mk
    (((1 - z) * n + z * h64) * mk).backward(dout.double())
    return m64.grad, h64.grad, W1d.grad, W2d.grad, b1d.grad, b2d.grad


@pytest.mark.parametrize("H", [64, 128, 256])
@pytest.mark.parametrize("profile", ["unit", "x1e-6", "x1e-20", "x1e+8", "rising", "falling", "one_hot_tile", "small_x_tiles",
                                     "rows_1e6_in_tile", "rows_1e8_in_tile"])
def test_gru_backward_range_guards(dev, profile, H):
    """The width-64, -128 and -256 GRU backward run on two fp16 pieces per operand behind power-of-two range guards (per
    32-atom tile for the gate gradients, per block and running for m | h; csrc/gru_bwd_f16.hip, gru_bwd128_f16.hip).  Gradient magnitudes that sit far
    from 1, that climb or fall by 1e8 across the batch (the running scale of the dW accumulators changes on the way),
    or that differ by 1e6 between neighbouring tiles (and m | h rows that differ by 1e3) must keep the float32 bar, per tensor AND per row of dm / dh.
    rows_*_in_tile put neighbouring rows of EVERY tile 1e6 / 1e8 apart: the gate gradients carry one scale per atom (a row
    scale factors out of dm | dh; the dW contraction folds it into that atom's m | h scale), so every row keeps the bar."""
    from mpnn_amd import ops
    V = 70_001                                               # 2188 tiles over 128-256 blocks: 9-17 tiles per block, ragged tail
    g = torch.Generator(device=dev).manual_seed(11)
    m, h, dout = (torch.randn(V, H, device=dev, generator=g) for _ in range(3))
    mask = (torch.rand(V, device=dev, generator=g) > 0.1).float()
    W1, W2 = (torch.randn(H, 3 * H, device=dev, generator=g) / 8 for _ in range(2))
    b1, b2 = (torch.randn(3 * H, device=dev, generator=g) / 8 for _ in range(2))
    ramp = torch.logspace(-8, 0, V, device=dev).reshape(-1, 1)
    if profile == "x1e-6":
        dout = dout * 1e-6
    elif profile == "x1e-20":
        dout = dout * 1e-20
    elif profile == "x1e+8":
        dout = dout * 1e8
    elif profile == "rising":
        dout = dout * ramp
    elif profile == "falling":
        dout = dout * ramp.flip(0)
    elif profile == "one_hot_tile":                           # every 7th tile carries gradients 1e6 larger
        tile = torch.arange(V, device=dev) // 32
        dout = dout * torch.where(tile % 7 == 0, 1e6, 1.0).reshape(-1, 1)
    elif profile in ("rows_1e6_in_tile", "rows_1e8_in_tile"):  # every other row of EVERY tile carries tiny gradients
        f = 1e-6 if profile == "rows_1e6_in_tile" else 1e-8
        odd = (torch.arange(V, device=dev) % 2 == 1)
        dout = dout * torch.where(odd, f, 1.0).reshape(-1, 1)
    elif profile == "small_x_tiles":                          # m | h of two tiles in three are 1e-3 of the others'
        tile = torch.arange(V, device=dev) // 32
        f = torch.where(tile % 3 == 0, 1.0, 1e-3).reshape(-1, 1)
        m, h = (m * f).contiguous(), (h * f).contiguous()
    dout = dout.contiguous()
    _, saved = ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)
    got = ops.gru_update_bwd_raw(dout, m, h, mask, W1, W2, saved)
    want = _gru_bwd_ref64(m, h, mask, dout, W1, W2, b1, b2)
    for a, b, name in zip(got, want, ("dm", "dh", "dW_ih", "dW_hh", "db_ih", "db_hh")):
        err = float((a.double() - b).abs().max() / b.abs().max())
        assert err < 1e-5, (profile, name, err)
    for k, name in ((0, "dm"), (1, "dh")):
        e = (got[k].double() - want[k]).abs().amax(1)
        s = want[k].abs().amax(1)
        live = s > 0
        worst = float((e[live] / s[live]).max())
        bar = {"one_hot_tile": 1e-4}.get(profile, 2e-5)
        assert worst < bar, (profile, name, worst)
        assert float(got[k][~live].abs().max()) == 0.0       # masked atoms: exact zeros


@pytest.mark.parametrize("H", [64, 128, 256])
@pytest.mark.parametrize("V", [1, 31, 32, 33, 255, 256, 257, 2049])
def test_gru_forward_backward_at_tile_boundaries(dev, V, H):
    """Atom counts around the kernels' granularities (32-atom tiles, 256-row rounds of the streamed kernels, fewer tiles
    than blocks): forward and every gradient of the raw GRU calls against float64, the ragged tail rows included."""
    from mpnn_amd import ops
    g = torch.Generator(device=dev).manual_seed(1000 * H + V)
    m, h, dout = (torch.randn(V, H, device=dev, generator=g) for _ in range(3))
    mask = (torch.rand(V, device=dev, generator=g) > 0.2).float()
    W1, W2 = (torch.randn(H, 3 * H, device=dev, generator=g) / (2 * H ** 0.5) for _ in range(2))
    b1, b2 = (torch.randn(3 * H, device=dev, generator=g) / 8 for _ in range(2))
    out, saved = ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)
    got = ops.gru_update_bwd_raw(dout, m, h, mask, W1, W2, saved)
    want = _gru_bwd_ref64(m, h, mask, dout, W1, W2, b1, b2)
    gi, gh = m.double() @ W1.double() + b1.double(), h.double() @ W2.double() + b2.double()
    mk = mask.double().reshape(-1, 1)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mk
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mk
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mk
    assert float((out.double() - ((1 - z) * n + z * h.double()) * mk).abs().max()) < 1e-5
    for a, b, name in zip(got, want, ("dm", "dh", "dW_ih", "dW_hh", "db_ih", "db_hh")):
        err = float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize("nf,mf,K,V", [(8, 8, 4, 60), (22, 22, 5, 333), (64, 64, 4, 3000), (128, 128, 4, 700),
                                       (256, 256, 3, 300), (64, 32, 2, 500), (64, 64, 100, 900),
                                       (64, 64, 5000, 2500), (24, 40, 4500, 2400), (130, 70, 4200, 2300)])
@pytest.mark.parametrize("gated", [False, True])
def test_edge_message_backward(dev, nf, mf, K, V, gated):
    from mpnn_amd import ops
    from mpnn_amd.graph import MolGraph
    rng = np.random.default_rng(nf + mf + K + V)
    deg = rng.integers(0, 6, V)
    row_ptr = np.zeros(V + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    E = int(row_ptr[-1])
    col = rng.integers(0, V, E).astype(np.int32)
    et = rng.integers(0, K, E).astype(np.int32)
    h = torch.from_numpy(rng.standard_normal((V, nf)).astype(np.float32))
    A = torch.from_numpy((rng.standard_normal((K, mf, nf)) / np.sqrt(nf)).astype(np.float32))
    gate = torch.from_numpy(rng.random((E, nf)).astype(np.float32)) if gated else None
    cot = torch.from_numpy(rng.standard_normal((E, mf)).astype(np.float32))
    # CPU reference in float64
    hd, Ad = h.double().requires_grad_(True), A.double().requires_grad_(True)
    gd = gate.double().requires_grad_(True) if gated else None
    x = hd[col.astype(np.int64)] * (gd if gated else 1.0)
    ref = torch.einsum("emn,en->em", Ad[et.astype(np.int64)], x)
    gref = torch.autograd.grad((ref * cot.double()).sum(), [hd, Ad] + ([gd] if gated else []))
    t = lambda a: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(col), None, t(et), torch.zeros(K, 1, device=dev),
                 torch.tensor([0, V], dtype=torch.int32, device=dev))
    hg, Ag = h.to(dev).requires_grad_(True), A.to(dev).requires_grad_(True)
    gg = gate.to(dev).requires_grad_(True) if gated else None
    out = ops.edge_message(hg, Ag, g, gate=gg)
    ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), [hg, Ag] + ([gg] if gated else []))
    for a, b, name in zip(ggpu, gref, ("dh", "dA", "dgate")):
        assert _rel(a.cpu(), b) < 2e-5, (name, _rel(a.cpu(), b))


@pytest.mark.parametrize("tag,nf,ef", [("h8_rand", 8, 4), ("h22_rand", 22, 7), ("h8_cont", 8, 4)])
def test_edge_network_backward(dev, tag, nf, ef):
    """Gradients flow through the HIP message + aggregator AND back into the tower parameters
    (dA from the kernel, tower by torch autograd on the K distinct rows)."""
    from mpnn_amd.mpnn_functions import AdjMsgAgg, EdgeNetwork
    # (a) per-pair form aggregated over real edges -- what BasicModel consumes
    f = Fixture("edge_network_%s_pairagg" % tag)
    m = EdgeNetwork(nf, ef, nf).to(dev)
    m.load_state_dict(f.params)
    m.pairwise = True
    afm = f.inputs["afm"].to(dev).requires_grad_(True)
    out = AdjMsgAgg(9)(m(afm, f.inputs["bfm"].to(dev)), f.inputs["adj"].to(dev))
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(afm.grad.cpu(), f.gin["afm"]) < TOL
    _check_param_grads(m, f, 2e-5)
    # (b) HEAD-fused form (non-member pairs and message_bias included)
    f = Fixture("edge_network_" + tag)
    m = EdgeNetwork(nf, ef, nf).to(dev)
    m.load_state_dict(f.params)
    afm = f.inputs["afm"].to(dev).requires_grad_(True)
    out = m(afm, f.inputs["bfm"].to(dev))
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(afm.grad.cpu(), f.gin["afm"]) < 2e-5
    _check_param_grads(m, f, 5e-5)


@pytest.mark.parametrize("tag,nf,ef", [("h8", 8, 4), ("h22", 22, 7)])
def test_att_edge_network_backward(dev, tag, nf, ef):
    from mpnn_amd.mpnn_functions import AdjMsgAgg, AttEdgeNetwork
    f = Fixture("att_edge_network_" + tag)
    m = AttEdgeNetwork(nf, ef, nf).to(dev)
    m.load_state_dict(f.params)
    afm = f.inputs["afm"].to(dev).requires_grad_(True)
    out = AdjMsgAgg(9)(m(afm, f.inputs["bfm"].to(dev)), f.inputs["adj"].to(dev))
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(afm.grad.cpu(), f.gin["afm"]) < TOL
    _check_param_grads(m, f, 2e-5)


@pytest.mark.parametrize("tag,H,ef", [("h8", 8, 4), ("h22", 22, 7)])
def test_basic_model_backward(dev, tag, H, ef):
    from mpnn_amd.models.basic_model import BasicModel
    from mpnn_amd.models.graph_model_wrapper import GraphWrapper
    f = Fixture("model_basic_" + tag)
    model = GraphWrapper(BasicModel(H, ef, H, 9, 6, message_opts={}, agg_opts={}, update_opts={},
                                    readout_opts={})).to(dev)
    model.load_state_dict(f.params)
    batch = {k: v.to(dev) for k, v in f.inputs.items()}
    batch["afm"].requires_grad_(True)
    out = model(batch)
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(batch["afm"].grad.cpu(), f.gin["afm"]) < 2e-5
    _check_param_grads(model, f, 5e-5)


def test_lipo_model_backward(dev):
    from mpnn_amd.models.graph_norm_wrapper import GraphWrapper
    from mpnn_amd.models.lipo_basic_model import BasicModel
    f = Fixture("model_lipo_T3_train")
    model = GraphWrapper(BasicModel(22, 7, 22, 9, 38, message_opts={}, agg_opts={}, update_opts={},
                                    readout_opts={}, message_steps=3), 3).to(dev)
    sd = dict(f.params)
    sd.update(f.pre)
    model.load_state_dict(sd)
    model.train()
    out = model({k: v.to(dev) for k, v in f.inputs.items()})
    (out * f.cot.to(dev)).sum().backward()
    _check_param_grads(model, f, 2e-4)      # six chained batch norms in the backward chain


@pytest.mark.parametrize("H,K,V", [(64, 4, 2500), (128, 4, 900), (256, 3, 500), (64, 1, 300), (32, 3, 400), (22, 5, 333)])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("need_dh", [False, True])
def test_message_aggregate_node(dev, H, K, V, weighted, need_dh):
    """ops.message_aggregate (message + adjacency-weighted sum as ONE autograd node): forward and the
    gradients of A (always) and h (optional: selects the fused dA-from-dagg path when absent) against
    a float64 reference -- this is the path BasicModel takes at every width."""
    from mpnn_amd import ops
    from mpnn_amd.graph import MolGraph
    rng = np.random.default_rng(H + K + V + weighted)
    deg = rng.integers(0, 6, V)
    row_ptr = np.zeros(V + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    E = int(row_ptr[-1])
    col = rng.integers(0, V, E).astype(np.int32)
    et = rng.integers(0, K, E).astype(np.int32)
    h = torch.from_numpy(rng.standard_normal((V, H)).astype(np.float32))
    A = torch.from_numpy((rng.standard_normal((K, H, H)) / np.sqrt(H)).astype(np.float32))
    w = torch.from_numpy((rng.random(E) + 0.5).astype(np.float32)) if weighted else None
    cot = torch.from_numpy(rng.standard_normal((V, H)).astype(np.float32))
    dst = np.repeat(np.arange(V), deg)
    hd, Ad = h.double().requires_grad_(need_dh), A.double().requires_grad_(True)
    msg = torch.einsum("emn,en->em", Ad[et.astype(np.int64)], hd[col.astype(np.int64)])
    if weighted:
        msg = msg * w.double().unsqueeze(1)
    ref = torch.zeros(V, H, dtype=torch.float64).index_add(0, torch.from_numpy(dst), msg)
    gref = torch.autograd.grad((ref * cot.double()).sum(), [Ad] + ([hd] if need_dh else []))
    t = lambda a: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(col), None, t(et), torch.zeros(K, 1, device=dev),
                 torch.tensor([0, V], dtype=torch.int32, device=dev))
    hg, Ag = h.to(dev).requires_grad_(need_dh), A.to(dev).requires_grad_(True)
    out = ops.message_aggregate(hg, Ag, g, w.to(dev) if weighted else None)
    ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), [Ag] + ([hg] if need_dh else []))
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-5
    for a, b, name in zip(ggpu, gref, ("dA", "dh")):
        assert _rel(a.cpu(), b) < 2e-5, (name, _rel(a.cpu(), b))


def test_ggnn_backward(dev):
    from mpnn_amd.mpnn_functions import GGNNMsgPass
    f = Fixture("ggnn_msg_pass")
    m = GGNNMsgPass(8, 4, 8).to(dev)
    m.load_state_dict(f.params)
    afm = f.inputs["afm"].to(dev).requires_grad_(True)
    out = m(afm, f.inputs["ibfm"].to(dev))
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(afm.grad.cpu(), f.gin["afm"]) < TOL
    _check_param_grads(m, f, 2e-5)


@pytest.mark.parametrize("name", ["agg_adj", "agg_adj_weighted", "agg_wadj", "agg_att_default", "agg_att_sigmoid"])
def test_aggregator_backward_on_dense_messages(dev, name):
    from torch import nn
    from mpnn_amd.mpnn_functions import AdjMsgAgg, AttMsgAgg, WAdjMsgAgg
    make = {"agg_adj": lambda: AdjMsgAgg(9), "agg_adj_weighted": lambda: AdjMsgAgg(9), "agg_wadj": lambda: WAdjMsgAgg(9),
            "agg_att_default": lambda: AttMsgAgg(1), "agg_att_sigmoid": lambda: AttMsgAgg(1, attn_act=nn.Sigmoid())}[name]
    f = Fixture(name)
    m = make().to(dev)
    if f.params:
        m.load_state_dict(f.params)
    msgs = f.inputs["messages"].to(dev).requires_grad_(True)
    out = m(msgs, f.inputs["adj"].to(dev))
    (out * f.cot.to(dev)).sum().backward()
    assert _rel(msgs.grad.cpu(), f.gin["messages"]) < TOL


@pytest.mark.parametrize("kind", ["adj", "wadj", "att_sigmoid"])
def test_aggregators_on_sparse_messages_backward(dev, kind):
    """Gradients (wrt atom features and every EdgeNetwork parameter) of the sparse message + aggregator
    pipeline against autograd through the oracle's dense pipeline -- covers the lazy message node, the
    non-member correction terms and the padded-row softmax."""
    from torch import nn
    from oracle import dense_ref as O
    from mpnn_amd.mpnn_functions import AdjMsgAgg, AttMsgAgg, EdgeNetwork, WAdjMsgAgg
    f = Fixture("edge_network_h8_rand")
    m = EdgeNetwork(8, 4, 8).to(dev)
    m.load_state_dict(f.params)
    m.pairwise = True
    # oracle side (CPU, float32 autograd), parameters as ONE leaf per distinct tensor
    leaves = {}
    p = {}
    for k, v in f.params.items():
        key = (v.data_ptr(), tuple(v.shape))
        if key not in leaves:
            leaves[key] = v.clone().requires_grad_(True)
        p[k] = leaves[key]
    afm_c = f.inputs["afm"].clone().requires_grad_(True)
    pair = O.edge_network_pair(p, afm_c, f.inputs["bfm"])
    if kind == "adj":
        agg, ref = AdjMsgAgg(9), O.agg_adj(pair, f.inputs["adj"])
    elif kind == "wadj":
        agg, ref = WAdjMsgAgg(9), O.agg_wadj(pair, f.inputs["adj"])
    else:
        agg = AttMsgAgg(1, attn_act=nn.Sigmoid())
        with torch.no_grad():
            agg.att[0].weight.fill_(0.7)
            agg.att[0].bias.fill_(-0.2)
        ap = {"att.0.weight": agg.att[0].weight.detach().clone(), "att.0.bias": agg.att[0].bias.detach().clone()}
        ref = O.agg_att(ap, pair, f.inputs["adj"], torch.sigmoid)
    cot = f.cot
    (ref * cot).sum().backward()
    afm = f.inputs["afm"].to(dev).requires_grad_(True)
    out = agg.to(dev)(m(afm, f.inputs["bfm"].to(dev)), f.inputs["adj"].to(dev))
    (out * cot.to(dev)).sum().backward()
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-5
    assert _rel(afm.grad.cpu(), afm_c.grad) < 5e-5
    checked = 0
    for k, prm in m.named_parameters():
        ref_g = p[k].grad
        if ref_g is None:
            continue
        assert _rel(prm.grad.cpu(), ref_g) < 1e-4, k
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("H,K,V", [(64, 4, 2500), (128, 4, 900), (64, 1, 300), (32, 3, 400)])
@pytest.mark.parametrize("weighted", [False, True])
def test_gated_message_aggregate_gate_gradient(dev, H, K, V, weighted):
    """Gated message + adjacency-weighted sum with gradients wanted for the gate and A but not for the node features
    (the attention models): at widths 64 / 128 both come from dout[dst(e)] directly (mpnn_edge_message_agg_bwd_dgate_f32,
    _da_f32); other widths take the unfused route.  float64 reference."""
    from mpnn_amd import ops
    from mpnn_amd.graph import MolGraph
    rng = np.random.default_rng(H + K + V + 7 * weighted)
    deg = rng.integers(0, 6, V)
    row_ptr = np.zeros(V + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    E = int(row_ptr[-1])
    col = rng.integers(0, V, E).astype(np.int32)
    et = rng.integers(0, K, E).astype(np.int32)
    h = torch.from_numpy(rng.standard_normal((V, H)).astype(np.float32))
    A = torch.from_numpy((rng.standard_normal((K, H, H)) / np.sqrt(H)).astype(np.float32))
    gate = torch.from_numpy(rng.random((E, H)).astype(np.float32))
    w = torch.from_numpy((rng.random(E) + 0.5).astype(np.float32)) if weighted else None
    cot = torch.from_numpy(rng.standard_normal((V, H)).astype(np.float32))
    dst = np.repeat(np.arange(V), deg)
    Ad, gd = A.double().requires_grad_(True), gate.double().requires_grad_(True)
    msg = torch.einsum("emn,en->em", Ad[et.astype(np.int64)], gd * h.double()[col.astype(np.int64)])
    if weighted:
        msg = msg * w.double().unsqueeze(1)
    ref = torch.zeros(V, H, dtype=torch.float64).index_add(0, torch.from_numpy(dst), msg)
    gref = torch.autograd.grad((ref * cot.double()).sum(), [Ad, gd])
    t = lambda a: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(col), None, t(et), torch.zeros(K, 1, device=dev),
                 torch.tensor([0, V], dtype=torch.int32, device=dev))
    Ag, gg = A.to(dev).requires_grad_(True), gate.to(dev).requires_grad_(True)
    out = ops.message_aggregate(h.to(dev), Ag, g, w.to(dev) if weighted else None, gate=gg)
    ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), [Ag, gg])
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-5
    for a, b, name in zip(ggpu, gref, ("dA", "dgate")):
        assert _rel(a.cpu(), b) < 2e-5, (name, _rel(a.cpu(), b))
