#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run only in the build container (needs /root/reference; the GPU box has neither
the reference nor any need for this script):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is Python 2 era and cannot be imported as a package under Python 3
(implicit relative imports, mpnn_functions/__init__.py:1-4), but every file on the
hot path is Python-3-clean, so each is loaded *by file path* under the bare module
name its siblings expect.  This script carries no reference source: it only calls
the loaded classes and stores inputs / parameters / outputs / gradients as arrays.

Every fixture is one .npz:
    in.<name>      inputs
    p.<key>        parameters and buffers, keys as in the reference state_dict (tensors that several
                   keys share are stored under the first key; `alias` = "key=first;..." lists the rest)
    pre.<key>      (models) running statistics BEFORE the recorded forward
    out[.<name>]   forward result(s)
    g.in.<name>    d(sum(out * cot))/d(input)     (cot = fixed random cotangent, saved as `cot`)
    g.p.<key>      d(sum(out * cot))/d(parameter)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MPNN_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from mpnn_amd import synth  # noqa: E402  (numpy-only synthetic molecules)


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    R = types.SimpleNamespace()
    en = _load("edge_network", "mpnn_functions/message/edge_network.py")
    aen = _load("att_edge_network", "mpnn_functions/message/att_edge_network.py")
    ggnn = _load("ggnn_msg_pass", "mpnn_functions/message/ggnn_msg_pass.py")
    bil = _load("bilinear_edge_network", "mpnn_functions/message/bilinear_edge_network.py")
    adj = _load("adjacent_message_agg", "mpnn_functions/message_aggregators/adjacent_message_agg.py")
    wadj = _load("weighted_adjacent_message_agg",
                 "mpnn_functions/message_aggregators/weighted_adjacent_message_agg.py")
    att = _load("attention_message_agg", "mpnn_functions/message_aggregators/attention_message_agg.py")
    gru = _load("gru_update", "mpnn_functions/update/gru_update.py")
    glo = _load("graph_level_output", "mpnn_functions/readout/graph_level_output.py")
    mbn = _load("mask_batch_norm", "models/mask_batch_norm.py")
    R.EdgeNetwork = en.EdgeNetwork
    R.AttEdgeNetwork = aen.AttEdgeNetwork
    R.GGNNMsgPass = ggnn.GGNNMsgPass
    R.BiLiniearEdgeNetwork = bil.BiLiniearEdgeNetwork
    R.AdjMsgAgg = adj.AdjMsgAgg
    R.WAdjMsgAgg = wadj.WAdjMsgAgg
    R.AttMsgAgg = att.AttMsgAgg
    R.GRUUpdate = gru.GRUUpdate
    R.GRUCell = gru.GRUCell
    R.GraphLevelOutput = glo.GraphLevelOutput
    R.MaskBatchNorm = mbn.MaskBatchNorm
    R.MaskBatchNorm1d = mbn.MaskBatchNorm1d
    # the model files do `from mpnn_functions import *` and
    # `from mpnn_functions.message.ggnn_msg_pass import GGNNMsgPass`
    pkg = types.ModuleType("mpnn_functions")
    pkg.__path__ = []
    for k in ("EdgeNetwork", "AttEdgeNetwork", "BiLiniearEdgeNetwork", "AdjMsgAgg", "WAdjMsgAgg",
              "AttMsgAgg", "GRUUpdate", "GraphLevelOutput"):
        setattr(pkg, k, getattr(R, k))
    pkg.__all__ = [k for k in vars(pkg) if not k.startswith("_")]
    sys.modules["mpnn_functions"] = pkg
    sub = types.ModuleType("mpnn_functions.message")
    sub.__path__ = []
    sys.modules["mpnn_functions.message"] = sub
    sys.modules["mpnn_functions.message.ggnn_msg_pass"] = ggnn
    R.basic_model = _load("ref_basic_model", "models/basic_model.py")
    R.lipo_basic_model = _load("ref_lipo_basic_model", "models/lipo_basic_model.py")
    R.graph_model_wrapper = _load("ref_graph_model_wrapper", "models/graph_model_wrapper.py")
    R.graph_norm_wrapper = _load("ref_graph_norm_wrapper", "models/graph_norm_wrapper.py")
    # models/att_model.py:3-4 does `from mpnn_functions import *` (its constructor's default readout is the name
    # Set2Vec, whose module needs rdkit) and `from batch_norm_graph_wrapper import MaskBatchNorm`
    pkg.Set2Vec = R.GraphLevelOutput                  # placeholder for the default argument only: never constructed
    pkg.__all__.append("Set2Vec")
    _load("batch_norm_graph_wrapper", "models/batch_norm_graph_wrapper.py")
    R.att_model = _load("ref_att_model", "models/att_model.py")
    return R


# --------------------------------------------------------------------------- helpers
def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def randomise(module, seed, bias_scale=0.3):
    """Non-degenerate parameters: default init, then biases made non-zero (so A0 != 0)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        seen = set()
        for name, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            if "bias" in name:
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 2 * bias_scale)


def save(name, inputs, module, out, cot=None, extra=None, grads=True):
    d = {}
    for k, v in inputs.items():
        d["in." + k] = v.detach().numpy()
    if module is not None:
        # shared tensors (the 50 aliases of the tower's one Linear, edge_network.py:20) are stored once
        first, alias = {}, []
        for k, v in module.state_dict().items():
            key = (v.data_ptr(), tuple(v.shape))
            if key in first and v.numel() > 0:
                alias.append("%s=%s" % (k, first[key]))
                continue
            first[key] = k
            d["p." + k] = v.detach().numpy()
        d["alias"] = np.array(";".join(alias))
    outs = out if isinstance(out, dict) else {"": out}
    for k, v in outs.items():
        d["out" + ("." + k if k else "")] = v.detach().numpy()
    if grads:
        main = outs[""] if "" in outs else list(outs.values())[0]
        g = torch.Generator().manual_seed(1234)
        if cot is None:
            cot = torch.rand(main.shape, generator=g) - 0.5
        d["cot"] = cot.numpy()
        leaves, names = [], []
        for k, v in inputs.items():
            if v.requires_grad:
                leaves.append(v)
                names.append("g.in." + k)
        if module is not None:
            seen = set()
            for k, p in module.named_parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    leaves.append(p)
                    names.append("g.p." + k)
        gr = torch.autograd.grad((main * cot).sum(), leaves, allow_unused=True)
        for n_, g_ in zip(names, gr):
            if g_ is not None:
                d[n_] = g_.detach().numpy()
    if extra:
        d.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote %-40s %6.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def ragged_batch(nf, ef, seed, continuous=False, lipo=False):
    """4 small molecules incl. a single-atom one and one with a degree-0 atom, N<=9."""
    rng = np.random.default_rng(seed + 1)
    sizes = [9, 5, 1, 7]
    G, N = len(sizes), max(sizes)
    afm = np.zeros((G, N, nf), np.float32)
    bfm = np.zeros((G, N, N, ef), np.float32)
    adj = np.zeros((G, N, N), np.float32)
    mask = np.zeros((G, N, 1), np.float32)
    for g, n in enumerate(sizes):
        mask[g, :n] = 1
        if lipo:
            oh = nf - 3
            afm[g, np.arange(n), rng.integers(0, oh, n)] = 1
            afm[g, :n, oh:] = rng.random((n, 3), dtype=np.float32)
        else:
            afm[g, :n] = rng.random((n, nf), dtype=np.float32) * 2 - 1
        bonds = []
        last = n - 1 if g == 1 else n      # molecule 1 keeps its last atom isolated (degree 0)
        for i in range(1, last):
            bonds.append((i, int(rng.integers(0, i))))
        if n >= 7:
            bonds.append((n - 1, 0) if (n - 1, 0) not in bonds and n - 1 < last else (2, 0))
        for (a, b) in set(bonds):
            if a == b:
                continue
            if continuous:
                f = rng.random(ef, dtype=np.float32)
            else:
                f = np.zeros(ef, np.float32)
                f[int(rng.choice(ef, p=np.asarray(synth.BOND_TYPE_P[:ef]) / sum(synth.BOND_TYPE_P[:ef])
                                 if ef <= 4 else None))] = 1
            adj[g, a, b] = adj[g, b, a] = 1
            bfm[g, a, b] = bfm[g, b, a] = f
    return t(afm), t(bfm), t(adj), t(mask)


# --------------------------------------------------------------------------- fixtures
def fx_csr(R):
    afm, bfm, adj, mask = ragged_batch(8, 4, 11)
    nz = adj.nonzero()
    B, N = adj.shape[:2]
    rows = nz[:, 0] * N + nz[:, 1]
    row_ptr = np.zeros(B * N + 1, np.int32)
    np.cumsum(np.bincount(rows.numpy(), minlength=B * N), out=row_ptr[1:])
    col = (nz[:, 0] * N + nz[:, 2]).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "csr_ragged.npz"), adj=adj.numpy(), nonzero=nz.numpy().astype(np.int32),
                        row_ptr=row_ptr, col_idx=col)
    print("wrote csr_ragged.npz")


def fx_edge_network(R):
    for tag, nf, ef, seed, mode in (("h8_rand", 8, 4, 21, "rand"), ("h8_init", 8, 4, 22, "init"),
                                    ("h22_rand", 22, 7, 23, "rand"), ("h8_cont", 8, 4, 24, "cont")):
        torch.manual_seed(seed)
        afm, bfm, adj, mask = ragged_batch(nf, ef, seed, continuous=(mode == "cont"))
        m = R.EdgeNetwork(nf, ef, nf)
        if mode == "init":
            m.apply(R.lipo_basic_model.BasicModel.init_weights)
        else:
            randomise(m, seed)
        with torch.no_grad():
            m.message_bias.copy_(torch.rand(nf) - 0.5)
        afm.requires_grad_(True)
        # HEAD behaviour: message fused with the un-masked all-pairs sum (edge_network.py:42-51)
        fused = m(afm, bfm)
        # legacy per-pair contract (the commented-out lines edge_network.py:40,52), evaluated with
        # the reference module's own edge_map: A = edge_map(bfm) viewed (B,N,N,mf,nf); m_ij = A_ij h_j
        B, N = afm.shape[:2]
        A = m.edge_map(bfm.view(-1, ef)).view(B, N, N, nf, nf)
        pair = A.matmul(afm.unsqueeze(1).unsqueeze(-1)).squeeze(-1)
        A0 = m.edge_map(torch.zeros(1, ef)).view(nf, nf)
        save("edge_network_" + tag, {"afm": afm, "bfm": bfm, "adj": adj, "mask": mask}, m,
             {"": fused, "pair": pair, "A0": A0})
        # gradients of the per-pair form aggregated over real edges (what BasicModel consumes)
        agg = R.AdjMsgAgg(N)(pair, adj)
        save("edge_network_" + tag + "_pairagg", {"afm": afm, "bfm": bfm, "adj": adj, "mask": mask}, m,
             {"": agg})


def fx_att_edge_network(R):
    for tag, nf, ef, seed in (("h8", 8, 4, 31), ("h22", 22, 7, 32)):
        torch.manual_seed(seed)
        afm, bfm, adj, mask = ragged_batch(nf, ef, seed)
        m = R.AttEdgeNetwork(nf, ef, nf)
        randomise(m, seed)
        afm.requires_grad_(True)
        B, N = afm.shape[:2]
        # AttEdgeNetwork.forward expects the legacy 5-D edge_embed (att_edge_network.py:31); set it
        # from the module's own edge_map and ask forward to reuse it.
        m.edge_embed = m.edge_map(bfm.view(-1, ef)).view(B, N, N, nf, nf)
        out = m(afm, bfm, reuse_graph_tensors=True)
        agg = R.AdjMsgAgg(N)(out, adj)
        save("att_edge_network_" + tag, {"afm": afm, "bfm": bfm, "adj": adj, "mask": mask}, m,
             {"": agg, "pair": out})


def fx_ggnn(R):
    torch.manual_seed(41)
    nf, ef = 8, 4
    afm, bfm, adj, mask = ragged_batch(nf, ef, 41)
    ibfm = (bfm.argmax(-1) + 1) * (bfm.sum(-1) > 0).long()     # 0 = no bond
    m = R.GGNNMsgPass(nf, ef, nf)
    m.init_weights()
    with torch.no_grad():
        m.message_bias.copy_(torch.rand(nf) - 0.5)
    afm.requires_grad_(True)
    out = m(afm, ibfm)
    save("ggnn_msg_pass", {"afm": afm, "ibfm": ibfm, "adj": adj, "mask": mask}, m, out)


def fx_bilinear(R):
    torch.manual_seed(45)
    nf = 3
    g = torch.Generator().manual_seed(45)
    afm = torch.rand(2, 4, nf, generator=g) - 0.5
    bfm = torch.rand(2, 4, 4, nf ** 3, generator=g) - 0.5
    afm.requires_grad_(True)
    m = R.BiLiniearEdgeNetwork(nf, nf ** 3, nf)
    out = m(afm, bfm)
    save("bilinear_edge_network", {"afm": afm, "bfm": bfm}, None, out)


def fx_aggregators(R):
    torch.manual_seed(51)
    afm, bfm, adj, mask = ragged_batch(8, 4, 51)
    B, N = adj.shape[:2]
    g = torch.Generator().manual_seed(51)
    msgs = (torch.rand(B, N, N, 8, generator=g) - 0.5).requires_grad_(True)
    save("agg_adj", {"messages": msgs, "adj": adj}, None, R.AdjMsgAgg(N)(msgs, adj))
    save("agg_wadj", {"messages": msgs, "adj": adj}, None, R.WAdjMsgAgg(N)(msgs, adj))
    m = R.AttMsgAgg(1)
    save("agg_att_default", {"messages": msgs, "adj": adj}, m, m(msgs, adj))
    m = R.AttMsgAgg(1, attn_act=nn.Sigmoid())
    with torch.no_grad():
        m.att[0].weight.fill_(0.7)
        m.att[0].bias.fill_(-0.2)
    save("agg_att_sigmoid", {"messages": msgs, "adj": adj}, m, m(msgs, adj))
    # weighted adjacency (adj values are used as multipliers, adjacent_message_agg.py:18)
    wadj = adj * (torch.rand(adj.shape, generator=g) + 0.5)
    wadj = (wadj + wadj.transpose(1, 2)) / 2
    save("agg_adj_weighted", {"messages": msgs, "adj": wadj}, None, R.AdjMsgAgg(N)(msgs, wadj))


def fx_gru(R):
    for tag, H, seed in (("h8", 8, 61), ("h22", 22, 62), ("h64", 64, 63)):
        torch.manual_seed(seed)
        afm, bfm, adj, mask = ragged_batch(H, 4, seed)
        g = torch.Generator().manual_seed(seed)
        msg = (torch.rand(afm.shape, generator=g) * 2 - 1).requires_grad_(True)
        h = (afm.clone() + 0.1 * (torch.rand(afm.shape, generator=g) - 0.5)).requires_grad_(True)
        m = R.GRUUpdate(H, H)
        randomise(m, seed)
        save("gru_update_" + tag, {"messages": msg, "node_states": h, "mask": mask}, m, m(msg, h, mask))


def fx_mask_bn(R):
    torch.manual_seed(71)
    afm, bfm, adj, mask = ragged_batch(8, 4, 71)
    x = afm.clone().requires_grad_(True)
    m = R.MaskBatchNorm1d(8)
    with torch.no_grad():
        m.weight.copy_(torch.rand(8) + 0.5)
        m.bias.copy_(torch.rand(8) - 0.5)
    m.train()
    y = m(x, mask)
    save("mask_bn1d_train", {"x": x, "mask": mask}, None, y,
         extra={"p.weight": m.weight.detach().numpy(), "p.bias": m.bias.detach().numpy(),
                "running_mean_after": m.running_mean.numpy(), "running_var_after": m.running_var.numpy()})
    m.eval()
    y = m(x, mask)
    save("mask_bn1d_eval", {"x": x, "mask": mask}, m, y)
    save("mask_bn_noaffine", {"x": x, "mask": mask}, None, R.MaskBatchNorm()(x, mask))


def fx_readout(R):
    torch.manual_seed(81)
    afm, bfm, adj, mask = ragged_batch(8, 4, 81)
    g = torch.Generator().manual_seed(81)
    x = (torch.rand(afm.shape[0], afm.shape[1], 16, generator=g) - 0.5).requires_grad_(True)
    m = R.GraphLevelOutput(8, 6)
    randomise(m, 81)
    save("graph_level_output_masked", {"x": x, "mask": mask}, m, m(x, mask=mask))
    save("graph_level_output_nomask", {"x": x}, m, m(x))


def fx_models(R):
    # (1) the model that runs at HEAD: lipo_basic_model under graph_norm_wrapper (test_lipo.py:123-129)
    for T in (3, 6):
        for mode in ("train", "eval"):
            torch.manual_seed(317)
            af, naf, ef = 19, 3, 7
            afm, bfm, adj, mask = ragged_batch(af + naf, ef, 91, lipo=True)
            batch = {"afm": afm[..., :af].contiguous(), "nafm": afm[..., af:].contiguous(),
                     "bfm": bfm, "adj": adj, "mask": mask}
            gm = R.lipo_basic_model.BasicModel(af + naf, ef, af + naf, adj.shape[-1], 2 * af,
                                               message_opts={}, agg_opts={}, update_opts={},
                                               readout_opts={}, message_steps=T)
            model = R.graph_norm_wrapper.GraphWrapper(gm, naf)
            model.apply(R.lipo_basic_model.BasicModel.init_weights)
            randomise(model, 92, bias_scale=0.05)
            if mode == "eval":
                model.train()
                with torch.no_grad():
                    model(batch)          # one training pass so running stats are non-trivial
                model.eval()
            else:
                model.train()
            pre = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}
            out = model(batch)
            d_in = {k: v for k, v in batch.items()}
            save("model_lipo_T%d_%s" % (T, mode), d_in, model, out,
                 extra={"pre." + k: v.numpy() for k, v in pre.items()})
    # (2) the intended north-star composition: basic_model.BasicModel.forward (basic_model.py:50-58)
    # with the per-pair message contract; at HEAD the stock forward raises, so the loop
    # is driven here with the reference's own sub-modules taken from a real BasicModel instance.
    for tag, H, ef, T, seed in (("h8", 8, 4, 3, 101), ("h22", 22, 7, 3, 102)):
        torch.manual_seed(seed)
        afm, bfm, adj, mask = ragged_batch(H, ef, seed)
        B, N = afm.shape[:2]
        bm = R.basic_model.BasicModel(H, ef, H, N, 6, message_opts={}, agg_opts={}, update_opts={},
                                      readout_opts={}, message_steps=T)
        model = R.graph_model_wrapper.GraphWrapper(bm)
        randomise(model, seed)
        afm.requires_grad_(True)
        A = bm.mf.edge_map(bfm.view(-1, ef)).view(B, N, N, H, H)
        node_state = afm
        states = []
        for i in range(bm.iters):
            pair = A.matmul(afm.unsqueeze(1).unsqueeze(-1)).squeeze(-1)
            node_state = bm.uf(bm.ma(pair, adj), node_state, mask)
            states.append(node_state)
        out = bm.of(torch.cat([node_state, afm], dim=-1), mask=mask)
        save("model_basic_" + tag, {"afm": afm, "bfm": bfm, "adj": adj, "mask": mask}, model,
             {"": out, "node_state": node_state, "state1": states[0]})


def fx_att_models(R):
    """(3) the configuration-3 model, models/att_model.py:55-59: one AttEdgeNetwork per step, AdjMsgAgg, GRU, the
    parameter-free MaskBatchNorm, readout on cat[state, afm].  Its stock forward raises at HEAD (AttEdgeNetwork wants
    the legacy 5-D edge_embed, att_edge_network.py:31, which _precompute_edge_embed no longer produces), so the loop
    is driven with the model's own sub-modules mfs[i], ma, uf, bn, of -- each mf with its edge_embed set from its own
    edge_map in the legacy layout and asked to reuse it, exactly as fx_att_edge_network does for the operator."""
    for tag, H, ef, T, seed in (("h8_T3", 8, 4, 3, 111), ("h22_T5", 22, 7, 5, 112)):
        torch.manual_seed(seed)
        afm, bfm, adj, mask = ragged_batch(H, ef, seed)
        B, N = afm.shape[:2]
        am = R.att_model.BasicModel(H, ef, H, N, 6, message_opts={}, agg_opts={}, update_opts={},
                                    readout_func=R.GraphLevelOutput, readout_opts={}, message_steps=T)
        randomise(am, seed)
        am.train()
        afm.requires_grad_(True)
        node_state = afm
        states = []
        for mf in am.mfs:
            mf.edge_embed = mf.edge_map(bfm.view(-1, ef)).view(B, N, N, H, H)
            pair = mf(afm, bfm, reuse_graph_tensors=True)
            node_state = am.bn(am.uf(am.ma(pair, adj), node_state, mask), mask)
            states.append(node_state)
        out = am.of(torch.cat([node_state, afm], dim=-1), mask=mask)
        save("model_att_" + tag, {"afm": afm, "bfm": bfm, "adj": adj, "mask": mask}, am,
             {"": out, "node_state": node_state, "state1": states[0]})


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not found at %s: golden vectors can only be regenerated in the build "
                 "container" % REF)
    torch.set_num_threads(1)
    R = load_reference()
    for fx in (fx_csr, fx_edge_network, fx_att_edge_network, fx_ggnn, fx_bilinear, fx_aggregators,
               fx_gru, fx_mask_bn, fx_readout, fx_models, fx_att_models):
        fx(R)


if __name__ == "__main__":
    main()
