"""The oracle (oracle/dense_ref.py) against vectors produced by the real reference.

CPU only.  Tolerance 2e-6 absolute on O(1) values: both sides are torch-CPU fp32, the only
difference is summation order inside einsum / bmm.
"""
import pytest
import torch

from conftest import Fixture, max_err
from oracle import dense_ref as O

TOL = 2e-6


def _leaf(t):
    return t.clone().requires_grad_(True)


def _grad_check(out, cot, leaves, expect, tol=5e-6):
    got = torch.autograd.grad((out * cot).sum(), list(leaves.values()), allow_unused=True)
    for (k, _), g in zip(leaves.items(), got):
        if k in expect:
            assert g is not None, k
            assert max_err(g, expect[k]) < tol * max(1.0, float(expect[k].abs().max())), k


@pytest.mark.parametrize("tag", ["h8_rand", "h8_init", "h22_rand", "h8_cont"])
def test_edge_network(tag):
    f = Fixture("edge_network_" + tag)
    afm = _leaf(f.inputs["afm"])
    p = {k: _leaf(v) for k, v in f.params.items()}
    # shared tower weight must stay ONE leaf so its gradient accumulates like the reference's
    alias = str(f.raw["alias"])
    for item in filter(None, alias.split(";")):
        k, first = item.split("=")
        p[k] = p[first]
    fused = O.edge_network_fused(p, afm, f.inputs["bfm"])
    assert max_err(fused, f.out[""]) < TOL * 10
    assert max_err(O.edge_network_pair(p, afm, f.inputs["bfm"]), f.out["pair"]) < TOL
    leaves = {"afm": afm}
    leaves.update({k: v for k, v in p.items() if k in f.gp})
    _grad_check(fused, f.cot, leaves, dict(afm=f.gin["afm"], **f.gp), tol=2e-5)
    if tag == "h8_init":
        assert float(f.out["A0"].abs().max()) == 0.0     # zero biases => edge_map(0) == 0
    else:
        assert float(f.out["A0"].abs().max()) > 1e-3


@pytest.mark.parametrize("tag", ["h8_rand", "h22_rand", "h8_cont"])
def test_edge_network_pair_aggregated(tag):
    f = Fixture("edge_network_%s_pairagg" % tag)
    afm = _leaf(f.inputs["afm"])
    out = O.agg_adj(O.edge_network_pair(f.params, afm, f.inputs["bfm"]), f.inputs["adj"])
    assert max_err(out, f.out[""]) < TOL
    _grad_check(out, f.cot, {"afm": afm}, {"afm": f.gin["afm"]})


@pytest.mark.parametrize("tag", ["h8", "h22"])
def test_att_edge_network(tag):
    f = Fixture("att_edge_network_" + tag)
    afm = _leaf(f.inputs["afm"])
    pair = O.att_edge_network_pair(f.params, afm, f.inputs["bfm"])
    assert max_err(pair, f.out["pair"]) < TOL
    out = O.agg_adj(pair, f.inputs["adj"])
    assert max_err(out, f.out[""]) < TOL
    _grad_check(out, f.cot, {"afm": afm}, {"afm": f.gin["afm"]})


def test_ggnn():
    f = Fixture("ggnn_msg_pass")
    afm = _leaf(f.inputs["afm"])
    out = O.ggnn_fused(f.params, afm, f.inputs["ibfm"])
    assert max_err(out, f.out[""]) < TOL
    _grad_check(out, f.cot, {"afm": afm}, {"afm": f.gin["afm"]})


def test_bilinear():
    f = Fixture("bilinear_edge_network")
    assert max_err(O.bilinear_pair(f.inputs["afm"], f.inputs["bfm"]), f.out[""]) < TOL


def test_aggregators():
    f = Fixture("agg_adj")
    assert max_err(O.agg_adj(f.inputs["messages"], f.inputs["adj"]), f.out[""]) < TOL
    f = Fixture("agg_adj_weighted")
    assert max_err(O.agg_adj(f.inputs["messages"], f.inputs["adj"]), f.out[""]) < TOL
    f = Fixture("agg_wadj")
    assert max_err(O.agg_wadj(f.inputs["messages"], f.inputs["adj"]), f.out[""]) < TOL
    f = Fixture("agg_att_default")
    out = O.agg_att(f.params, f.inputs["messages"], f.inputs["adj"])
    assert max_err(out, f.out[""]) < TOL
    # default attention == softmax over a size-1 axis == plain all-pairs sum
    assert max_err(out, f.inputs["messages"].sum(dim=-2)) < TOL
    f = Fixture("agg_att_sigmoid")
    assert max_err(O.agg_att(f.params, f.inputs["messages"], f.inputs["adj"], torch.sigmoid), f.out[""]) < TOL


@pytest.mark.parametrize("tag", ["h8", "h22", "h64"])
def test_gru_update(tag):
    f = Fixture("gru_update_" + tag)
    m, h = _leaf(f.inputs["messages"]), _leaf(f.inputs["node_states"])
    p = {k: _leaf(v) for k, v in f.params.items()}
    out = O.gru_update(p, m, h, f.inputs["mask"])
    assert max_err(out, f.out[""]) < TOL
    mask = f.inputs["mask"]
    assert float((out * (1 - mask)).abs().max()) == 0.0      # padded rows exactly zero
    leaves = {"messages": m, "node_states": h}
    leaves.update(p)
    _grad_check(out, f.cot, leaves, dict(messages=f.gin["messages"], node_states=f.gin["node_states"], **f.gp))


def test_mask_batch_norm():
    f = Fixture("mask_bn1d_train")
    rm, rv = torch.zeros(8), torch.ones(8)
    y, rm2, rv2 = O.mask_bn1d(f.inputs["x"], f.inputs["mask"], f.params["weight"], f.params["bias"], rm, rv, True)
    assert max_err(y, f.out[""]) < TOL
    assert max_err(rm2, torch.from_numpy(f.raw["running_mean_after"])) < TOL
    assert max_err(rv2, torch.from_numpy(f.raw["running_var_after"])) < TOL
    f = Fixture("mask_bn1d_eval")
    y, _, _ = O.mask_bn1d(f.inputs["x"], f.inputs["mask"], f.params["weight"], f.params["bias"],
                          f.params["running_mean"], f.params["running_var"], False)
    assert max_err(y, f.out[""]) < TOL
    f = Fixture("mask_bn_noaffine")
    assert max_err(O.mask_bn(f.inputs["x"], f.inputs["mask"]), f.out[""]) < TOL


def test_graph_level_output():
    f = Fixture("graph_level_output_masked")
    assert max_err(O.graph_level_output(f.params, f.inputs["x"], f.inputs["mask"]), f.out[""]) < TOL
    f = Fixture("graph_level_output_nomask")
    assert max_err(O.graph_level_output(f.params, f.inputs["x"]), f.out[""]) < TOL


@pytest.mark.parametrize("tag", ["h8", "h22"])
def test_basic_model(tag):
    f = Fixture("model_basic_" + tag)
    afm = _leaf(f.inputs["afm"])
    gp = O.sub(f.params, "graph_model.")
    out, h = O.basic_model_forward(gp, afm, f.inputs["bfm"], f.inputs["adj"], f.inputs["mask"], 3, True)
    assert max_err(h, f.out["node_state"]) < TOL
    assert max_err(out, f.out[""]) < TOL * 5
    _grad_check(out, f.cot, {"afm": afm}, {"afm": f.gin["afm"]})


@pytest.mark.parametrize("tag,T", [("h8_T3", 3), ("h22_T5", 5)])
def test_att_model(tag, T):
    """The configuration-3 model (models/att_model.py:55-59) as the real reference modules compute it."""
    f = Fixture("model_att_" + tag)
    afm = _leaf(f.inputs["afm"])
    p = {k: _leaf(v) for k, v in f.params.items()}
    alias = str(f.raw["alias"])
    for item in filter(None, alias.split(";")):
        k, first = item.split("=")
        p[k] = p[first]
    out, h = O.att_model_forward(p, afm, f.inputs["bfm"], f.inputs["adj"], f.inputs["mask"], T, True)
    assert max_err(h, f.out["node_state"]) < 1e-5          # T chained parameter-free norms
    assert max_err(out, f.out[""]) < 1e-5
    leaves = {"afm": afm}
    leaves.update({k: v for k, v in p.items() if k in f.gp})
    _grad_check(out, f.cot, leaves, dict(afm=f.gin["afm"], **f.gp), tol=5e-5)


@pytest.mark.parametrize("T", [3, 6])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_lipo_model(T, mode):
    f = Fixture("model_lipo_T%d_%s" % (T, mode))
    params = dict(f.params)
    params.update(f.pre)                       # running stats as they were BEFORE the recorded forward
    out, buf = O.lipo_model_forward(params, f.inputs, steps=T, training=(mode == "train"), return_buffers=True)
    assert max_err(out, f.out[""]) < 2e-5      # 2*T chained batch-norms amplify rounding
    if mode == "train":
        for k, v in buf.items():
            assert max_err(v, f.params[k]) < 1e-5, k


def test_index_oracle():
    f = Fixture("csr_ragged")
    adj = torch.from_numpy(f.raw["adj"])
    rp, ci, w = O.dense_to_csr(adj)
    assert torch.equal(rp, torch.from_numpy(f.raw["row_ptr"]))
    assert torch.equal(ci, torch.from_numpy(f.raw["col_idx"]))


def test_set2vec_restatement_properties():
    """Set2Vec cannot be pinned by reference vectors (the module imports rdkit); check the properties its source
    text implies instead: the read is invariant to the order of atoms inside a molecule, padded atoms (mask 0)
    carry exactly zero attention, and the attention of a step sums to 1 over the WHOLE batch (softmax dim 0)."""
    import torch
    from oracle import dense_ref as O
    torch.manual_seed(3)
    B, N, nf = 3, 6, 8
    params = {"q_attn.weight": torch.randn(nf, nf) * 0.3, "e_attn.weight": torch.randn(1, nf) * 0.3}
    for g in "ifgo":
        params["lstmcell.w_h" + g] = torch.randn(2 * nf, nf) * 0.2
        params["lstmcell.b_h" + g] = torch.randn(1, nf) * 0.1
    mask = torch.ones(B, N, 1)
    mask[0, 4:] = 0
    mask[2, 3:] = 0
    x = torch.randn(B, N, nf) * mask
    out = O.set2vec(params, x, mask, steps=5)
    assert out.shape == (B, 2 * nf)
    perm = torch.tensor([2, 0, 3, 1, 4, 5])                   # permute real atoms of molecule 0 only
    x2 = x.clone()
    x2[0] = x[0, perm]
    assert (O.set2vec(params, x2, mask, steps=5) - out).abs().max() < 1e-6
    x3 = x.clone()
    x3[0, 4:] = 7.0                                            # garbage in padded slots must not matter
    assert (O.set2vec(params, x3, mask, steps=5) - out).abs().max() < 1e-6
    # one step by hand: m0 = 0 -> query; attention over all B*N atoms
    one = O.set2vec(params, x, mask, steps=1)
    i = torch.sigmoid(params["lstmcell.b_hi"]); g = torch.tanh(params["lstmcell.b_hg"]); o = torch.sigmoid(params["lstmcell.b_ho"])
    m = (o * torch.tanh(i * g)).expand(B, nf)
    q = m @ params["q_attn.weight"].t()
    e = torch.tanh(q.unsqueeze(1) + x).reshape(-1, nf) @ params["e_attn.weight"].t() + ((1 - mask) * -1e8).view(-1, 1)
    att = torch.softmax(e, dim=0).view(B, N, 1)
    assert abs(float(att.sum()) - 1.0) < 1e-6
    assert (one - torch.cat([m, (att * x).sum(1)], dim=1)).abs().max() < 1e-6
