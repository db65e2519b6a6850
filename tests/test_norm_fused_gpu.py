"""the masked batch norm of models/mask_batch_norm.py:5-38 fused into the GRU update
(mpnn_gru_update_norm_f32): the norm's moments come out of the update kernel's epilogue, the norm itself is applied
where the NEXT update reads its state.  Checked against the float64 composition of the reference formulas and against
the standalone kernels; the attention model at hidden 128 runs the fused loop against the oracle in
tests/test_configs_gpu.py::test_attention_model_at_c3_shape and at full size in tests/test_parity_round3_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    d = torch.device("cuda:0")
    from mpnn_amd import ops
    if not ops.gru_norm_applies(128, torch.zeros(1, device=d)):
        pytest.skip("the fused update + norm kernels are split-precision kernels (not under MPNN_GRU_MATH=fp32)")
    return d


def _gru64(m, h, mask, W_ih, W_hh, b_ih, b_hh):
    """gru_update.py:26-35,66-68 in float64."""
    H = h.shape[1]
    gi, gh = m @ W_ih + b_ih, h @ W_hh + b_hh
    mk = mask.unsqueeze(1)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mk
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mk
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mk
    return ((1 - z) * n + z * h) * mk


def _norm64(y, mask, weight, bias, eps, masked_mean, eps_inside):
    """mask_batch_norm.py:9-15 (parameter-free, eps inside the root) / :20-38 (affine, eps outside) in float64."""
    mk = mask.unsqueeze(1)
    cnt = mask.sum()
    mean = ((y * mk) if masked_mean else y).sum(0) / cnt
    var = (((y - mean) * mk) ** 2).sum(0) / cnt
    s = torch.sqrt(var + eps) if eps_inside else torch.sqrt(var) + eps
    out = (y - mean) / s
    if weight is not None:
        out = out * weight + bias
    return out * mk


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("H,V,affine,partial_mask", [(128, 1000, False, True), (128, 4096, True, False),
                                                     (256, 777, False, True), (256, 2048, True, True),
                                                     (128, 31, False, False)])
def test_two_updates_with_the_norm_between_them(dev, H, V, affine, partial_mask):
    """y1 = update(m1, h0); y2 = update(m2, norm(y1)); out = norm(y2): the fused form (moments from the epilogue, norm
    applied on the way into the next update, one apply pass at the end) against float64, forward and every gradient."""
    from mpnn_amd import ops
    torch.manual_seed(V + H)
    g = torch.Generator().manual_seed(3 * V + H)
    mk = (torch.rand(V, generator=g) > 0.2).float() if partial_mask else torch.ones(V)
    mk[0] = 1.0
    leaves = dict(m1=torch.randn(V, H, generator=g), m2=torch.randn(V, H, generator=g) * 0.7,
                  h0=torch.randn(V, H, generator=g) * mk.unsqueeze(1),
                  W_ih=torch.randn(H, 3 * H, generator=g) / H ** 0.5, W_hh=torch.randn(H, 3 * H, generator=g) / H ** 0.5,
                  b_ih=torch.randn(3 * H, generator=g) * 0.1, b_hh=torch.randn(3 * H, generator=g) * 0.1)
    if affine:
        leaves.update(gamma=torch.rand(H, generator=g) + 0.5, beta=torch.randn(H, generator=g) * 0.2)
        leaves["gamma"][:3] = torch.tensor([0.0, 1e-6, -1e-6])   # weight entries at / near zero keep exact gradients
    cot = torch.randn(V, H, generator=g)
    eps, masked_mean, eps_inside = (1e-5, True, False) if affine else (1e-6, False, True)
    flags = ops.BN_MASKED_MEAN if affine else ops.BN_EPS_INSIDE

    ref = {k: v.double().requires_grad_(True) for k, v in leaves.items()}
    w64, b64 = (ref["gamma"], ref["beta"]) if affine else (None, None)
    y1 = _gru64(ref["m1"], ref["h0"], mk.double(), ref["W_ih"], ref["W_hh"], ref["b_ih"], ref["b_hh"])
    y2 = _gru64(ref["m2"], _norm64(y1, mk.double(), w64, b64, eps, masked_mean, eps_inside), mk.double(), ref["W_ih"],
                ref["W_hh"], ref["b_ih"], ref["b_hh"])
    out64 = _norm64(y2, mk.double(), w64, b64, eps, masked_mean, eps_inside)
    (out64 * cot.double()).sum().backward()

    t = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
    mkd = mk.to(dev)
    w, b = (t["gamma"], t["beta"]) if affine else (None, None)
    a1, mom1 = ops.gru_update_norm_in(t["m1"], t["h0"], mkd, t["W_ih"], t["W_hh"], t["b_ih"], t["b_hh"])
    a2, mom2 = ops.gru_update_norm_in(t["m2"], a1, mkd, t["W_ih"], t["W_hh"], t["b_ih"], t["b_hh"], moments=mom1,
                                      weight=w, bias=b, eps=eps, flags=flags)
    out = ops.masked_batch_norm_given(a2, mkd, mom2, weight=w, bias=b, eps=eps, flags=flags)
    (out * cot.to(dev)).sum().backward()

    assert _rel(a1, y1.detach()) < 1e-5
    mean1 = (y1.detach() * mk.double().unsqueeze(1)).sum(0) / mk.sum()
    assert _rel(mom1.mean_var()[0], mean1) < 1e-5                                  # the epilogue's moments
    assert _rel(a2, y2.detach()) < 2e-5
    assert _rel(out, out64.detach()) < 2e-5
    assert float(out[mkd == 0].abs().max()) == 0.0 if partial_mask else True
    for k in leaves:
        assert _rel(t[k].grad, ref[k].grad) < 1e-4, (k, _rel(t[k].grad, ref[k].grad))

    # the whole chain as ONE autograd node: the norms' backward passes fused into the GRU backward kernels as well
    c = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
    w, b = (c["gamma"], c["beta"]) if affine else (None, None)
    chain = ops.gru_norm_chain(c["h0"], [c["m1"], c["m2"]], mkd, c["W_ih"], c["W_hh"], c["b_ih"], c["b_hh"], weight=w,
                               bias=b, eps=eps, flags=flags)
    (chain * cot.to(dev)).sum().backward()
    assert _rel(chain, out64.detach()) < 2e-5
    for k in leaves:
        assert _rel(c[k].grad, ref[k].grad) < 1e-4, ("chain", k, _rel(c[k].grad, ref[k].grad))
    with torch.no_grad():
        inf = ops.gru_norm_chain(c["h0"], [c["m1"], c["m2"]], mkd, c["W_ih"], c["W_hh"], c["b_ih"], c["b_hh"], weight=w,
                                 bias=b, eps=eps, flags=flags)
    assert torch.equal(inf, chain.detach())

    # the standalone kernels on the same inputs (what runs at every other width)
    s = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
    w, b = (s["gamma"], s["beta"]) if affine else (None, None)
    u1 = ops.gru_update(s["m1"], s["h0"], mkd, s["W_ih"], s["W_hh"], s["b_ih"], s["b_hh"])
    n1, _, _ = ops.masked_batch_norm(u1, mkd, w, b, None, eps, flags)
    u2 = ops.gru_update(s["m2"], n1, mkd, s["W_ih"], s["W_hh"], s["b_ih"], s["b_hh"])
    n2, _, _ = ops.masked_batch_norm(u2, mkd, w, b, None, eps, flags)
    (n2 * cot.to(dev)).sum().backward()
    assert _rel(out, n2.detach()) < 2e-5
    for k in leaves:
        assert _rel(t[k].grad, s[k].grad) < 1e-4, k


def test_attention_model_fused_norm_equals_the_standalone_norm_kernels(dev):
    """models/att_model.py:55-59 at hidden 128: fuse_norm on and off give the same node state, output and gradients."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.att_model import BasicModel as AttModel
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    H, T = 128, 4
    mb = synth.make_molecules(300, H, seed=11)
    torch.manual_seed(5)
    model = AttModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T,
                     readout_func=GraphLevelOutput).to(dev)
    g = MolGraph.from_molbatch(mb, dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    mask = torch.ones(afm.shape[0], 1, device=dev)
    res = []
    for fused in (True, False):
        model.fuse_norm = fused
        model.zero_grad()
        assert model._norm_fusable(afm) == fused
        out = model(afm, g, g, mask)
        out.square().sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert _rel(res[0][0], res[1][0]) < 2e-5
    assert len(res[0][1]) == len(res[1][1]) >= 20
    for k, gr in res[0][1].items():
        assert _rel(gr, res[1][1][k]) < 2e-4, (k, _rel(gr, res[1][1][k]))


def test_three_step_chain_with_small_gradients(dev):
    """Three updates (two fused norm backward passes in a row), cotangent of order 1e-6, hidden 256."""
    from mpnn_amd import ops
    H, V, T = 256, 1500, 3
    g = torch.Generator().manual_seed(9)
    mk = (torch.rand(V, generator=g) > 0.1).float()
    leaves = dict(h0=torch.randn(V, H, generator=g) * mk.unsqueeze(1), W_ih=torch.randn(H, 3 * H, generator=g) / H ** 0.5,
                  W_hh=torch.randn(H, 3 * H, generator=g) / H ** 0.5, b_ih=torch.randn(3 * H, generator=g) * 0.1,
                  b_hh=torch.randn(3 * H, generator=g) * 0.1)
    for t in range(T):
        leaves["m%d" % t] = torch.randn(V, H, generator=g) * (0.5 + t)
    cot = torch.randn(V, H, generator=g) * 1e-6
    ref = {k: v.double().requires_grad_(True) for k, v in leaves.items()}
    st = ref["h0"]
    for t in range(T):
        st = _norm64(_gru64(ref["m%d" % t], st, mk.double(), ref["W_ih"], ref["W_hh"], ref["b_ih"], ref["b_hh"]),
                     mk.double(), None, None, 1e-6, False, True)
    (st * cot.double()).sum().backward()
    c = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
    out = ops.gru_norm_chain(c["h0"], [c["m%d" % t] for t in range(T)], mk.to(dev), c["W_ih"], c["W_hh"], c["b_ih"], c["b_hh"])
    (out * cot.to(dev)).sum().backward()
    assert _rel(out, st.detach()) < 3e-5
    for k in leaves:
        assert _rel(c[k].grad, ref[k].grad) < 2e-4, (k, _rel(c[k].grad, ref[k].grad))


def test_norm_constant_kernels_against_torch(dev):
    """mpnn_norm_fold_f32 / mpnn_norm_bwd_consts_f32 against the same formulas in float64 torch ops."""
    from mpnn_amd import _lib, ops
    lib = _lib.load()
    H, n = 128, 1000.0
    g = torch.Generator().manual_seed(4)
    y = torch.randn(int(n), H, generator=g).double() * 0.7 + 0.3
    sums = torch.cat([y.sum(0), (y * y).sum(0)]).to(dev)
    count = torch.tensor([n], device=dev)
    gamma, beta = (torch.rand(H, generator=g) + 0.5).to(dev), (torch.randn(H, generator=g) * 0.2).to(dev)
    gamma[:4] = torch.tensor([0.0, 1e-6, -1e-6, 1e-12], device=dev)    # a zero weight keeps its gradient
    W, b = torch.randn(H, 3 * H, generator=g).to(dev), torch.randn(3 * H, generator=g).to(dev)
    for weight, bias, eps, flags in ((None, None, 1e-6, ops.BN_EPS_INSIDE), (gamma, beta, 1e-5, ops.BN_MASKED_MEAN)):
        mean, var, hs, ht, Wf, bf = ops._norm_fold(sums, count, weight, bias, W, b, eps, flags)
        mu = y.mean(0)
        v = (y * y).mean(0) - mu * mu
        s = torch.sqrt(v + eps) if flags & ops.BN_EPS_INSIDE else torch.sqrt(v) + eps
        g64 = weight.double().cpu() if weight is not None else torch.ones(H, dtype=torch.float64)
        b64 = bias.double().cpu() if bias is not None else torch.zeros(H, dtype=torch.float64)
        assert _rel(mean, mu) < 1e-6 and _rel(var, v) < 1e-6
        assert _rel(hs, g64 / s) < 1e-6 and _rel(ht, b64 - mu * g64 / s) < 1e-6
        assert _rel(Wf, W.double().cpu() * (g64 / s).unsqueeze(1)) < 1e-6
        assert _rel(bf, b.double().cpu() + (b64 - mu * g64 / s) @ W.double().cpu()) < 1e-5
        d = torch.randn(int(n), H, generator=g).double()
        bs = torch.cat([d.sum(0), (d * y).sum(0)]).to(dev)           # the sums run against the norm's RAW input
        kn = torch.empty(3 * H, device=dev)
        dw = torch.zeros(H, device=dev) if weight is not None else None
        db = torch.zeros(H, device=dev) if weight is not None else None
        _lib.check(lib.mpnn_norm_bwd_consts_f32(_lib.ptr(bs), _lib.fptr(mean), _lib.fptr(var), _lib.fptr(count),
                                                _lib.fptr(weight), _lib.fptr(kn), _lib.fptr(dw),
                                                _lib.fptr(db), H, eps, flags, _lib.stream()), "consts")
        yl = y.clone().requires_grad_(True)
        gl = g64.clone().requires_grad_(True)
        m2 = yl.mean(0)
        v2 = ((yl - m2) ** 2).mean(0)
        s2 = torch.sqrt(v2 + eps) if flags & ops.BN_EPS_INSIDE else torch.sqrt(v2) + eps
        (((yl - m2) / s2 * gl + b64) * d).sum().backward()
        k1, k2, k4 = kn[:H].double().cpu(), kn[H:2 * H].double().cpu(), kn[2 * H:].double().cpu()
        assert _rel(d * k1 + y * k2 + k4, yl.grad) < 2e-5
        if weight is not None:
            assert _rel(dw, gl.grad) < 1e-5 and float(dw[0].abs()) > 1e-3      # (the zero entry's gradient is not zero)
            assert _rel(db, d.sum(0)) < 1e-5


@pytest.mark.parametrize("H,V", [(22, 777), (38, 1200), (64, 500)])
@pytest.mark.parametrize("eval_mode", [False, True])
def test_generic_width_chain_like_the_lipo_model(dev, H, V, eval_mode):
    """Widths without a wide kernel run the fused update + norm on the generic fp32 kernel (what the lipo model's 22-38
    features use): affine norm with eps outside the root and masked mean (MaskBatchNorm1d, mask_batch_norm.py:20-38), batch
    statistics or given (running) ones, three updates, against float64."""
    from mpnn_amd import ops
    T = 3
    g = torch.Generator().manual_seed(V + H)
    mk = (torch.rand(V, generator=g) > 0.15).float()
    leaves = dict(h0=torch.randn(V, H, generator=g) * mk.unsqueeze(1), W_ih=torch.randn(H, 3 * H, generator=g) / H ** 0.5,
                  W_hh=torch.randn(H, 3 * H, generator=g) / H ** 0.5, b_ih=torch.randn(3 * H, generator=g) * 0.1,
                  b_hh=torch.randn(3 * H, generator=g) * 0.1, gamma=torch.rand(H, generator=g) + 0.5,
                  beta=torch.randn(H, generator=g) * 0.2)
    leaves["gamma"][:3] = torch.tensor([0.0, 1e-6, -1e-6])       # weight entries at / near zero keep exact gradients
    for t in range(T):
        leaves["m%d" % t] = torch.randn(V, H, generator=g)
    rmean, rvar = torch.randn(H, generator=g) * 0.1, torch.rand(H, generator=g) + 0.5
    cot = torch.randn(V, H, generator=g)
    eps = 1e-5
    ref = {k: v.double().requires_grad_(True) for k, v in leaves.items()}
    st = ref["h0"]
    want_stats = []
    for t in range(T):
        y = _gru64(ref["m%d" % t], st, mk.double(), ref["W_ih"], ref["W_hh"], ref["b_ih"], ref["b_hh"])
        if eval_mode:
            st = ((y - rmean.double()) / (rvar.double().sqrt() + eps) * ref["gamma"] + ref["beta"]) * mk.double().unsqueeze(1)
        else:
            mean = (y * mk.double().unsqueeze(1)).sum(0) / mk.sum()
            want_stats.append((mean.detach(), ((((y - mean) * mk.double().unsqueeze(1)) ** 2).sum(0) / mk.sum()).detach()))
            st = _norm64(y, mk.double(), ref["gamma"], ref["beta"], eps, True, False)
    (st * cot.double()).sum().backward()
    c = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
    out, stats = ops.gru_norm_chain(c["h0"], [c["m%d" % t] for t in range(T)], mk.to(dev), c["W_ih"], c["W_hh"], c["b_ih"],
                                    c["b_hh"], weight=c["gamma"], bias=c["beta"], eps=eps, flags=ops.BN_MASKED_MEAN,
                                    given=(rmean.to(dev), rvar.to(dev)) if eval_mode else None, return_stats=True)
    (out * cot.to(dev)).sum().backward()
    assert _rel(out, st.detach()) < 3e-5
    if not eval_mode:
        for (m_got, v_got), (m_want, v_want) in zip(stats, want_stats):
            assert _rel(m_got, m_want) < 1e-5 and _rel(v_got, v_want) < 1e-5
    for k in leaves:
        assert _rel(c[k].grad, ref[k].grad) < 2e-4, (k, _rel(c[k].grad, ref[k].grad))


def test_unsupported_width_is_refused():
    from mpnn_amd import _lib, ops
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    dev = torch.device("cuda:0")
    assert not ops.gru_norm_applies(64, torch.zeros(1, device=dev))       # no FAST fused kernel at 64 ...
    assert ops.gru_norm_kind(64, torch.zeros(1, device=dev)) == 1         # ... the generic one serves it
    assert not ops.gru_norm_costs_nothing(64, torch.zeros(1, device=dev)) and ops.gru_norm_costs_nothing(22, torch.zeros(1, device=dev))
    assert ops.gru_norm_kind(128, torch.zeros(1)) == 0                    # host tensors: none
    x = torch.zeros(8, 300, device=dev)
    with pytest.raises(_lib.MpnnError):
        ops.gru_update_norm_in(x, x, None, torch.zeros(300, 900, device=dev), torch.zeros(300, 900, device=dev),
                               torch.zeros(900, device=dev), torch.zeros(900, device=dev))
