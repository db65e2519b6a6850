"""ORACLE -- test infrastructure, not product code.

A dense, padded, CPU (torch float32) restatement of the reference's message-passing
path, written functionally over plain parameter dicts keyed like the reference's
state_dict.  It exists so that the HIP path can be checked on a box where the
reference itself cannot travel (the GPU box) and so that `bench.py` can time "the
reference CPU path" next to the GPU number.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Nothing under mpnn_amd/ does, and the product path has no CPU fallback.

Pinning: every function below is checked against golden vectors produced by the real
reference modules (tests/golden/make_golden.py imports them by path in the build
container); see tests/test_oracle_golden.py.  Parity is therefore pinned by
reference-generated fixtures (the reference ships no tests of its own).

All citations are relative to /root/reference.
"""
import re

import torch
import torch.nn.functional as F

_BIG_NEGATIVE = -1e8


# ----------------------------------------------------------------------------- params
def expand_aliases(params, alias):
    """Undo the fixture's de-duplication of shared tensors ("key=first;key=first...")."""
    out = dict(params)
    alias = str(alias)
    if alias:
        for item in alias.split(";"):
            k, first = item.split("=")
            out[k] = params[first]
    return out


def sub(params, prefix):
    """Entries of `params` under `prefix`, with the prefix stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}


def tower_plan(params, prefix="edge_map."):
    """Ordered layers of the edge tower as (weight, bias_or_None, followed_by_activation).

    Follows EdgeNetwork.__init__ (mpnn_functions/message/edge_network.py:14-26): widening
    Linear+act pairs, then 50 aliases of one bias-free Linear+act, then a last Linear with
    no activation.  The structure is read back from the state_dict keys:
    `<i>.weight/<i>.bias` = plain Linear, `<i>.0.weight` = the Sequential(Linear, act) alias.
    """
    idx = {}
    for k in params:
        if not k.startswith(prefix):
            continue
        m = re.match(r"(\d+)\.(0\.)?(weight|bias)$", k[len(prefix):])
        if m:
            idx.setdefault(int(m.group(1)), {})[(m.group(2) or "") + m.group(3)] = params[k]
    order = sorted(idx)
    plan = []
    for pos, i in enumerate(order):
        ent = idx[i]
        if "0.weight" in ent:
            plan.append((ent["0.weight"], None, True))
        else:
            plan.append((ent["weight"], ent.get("bias"), pos != len(order) - 1))
    return plan


def edge_map(params, x, act=F.relu, prefix="edge_map."):
    """edge_map(x) for rows x (R, ef) -> (R, mf*nf); edge_network.py:14-26,36-37."""
    for w, b, has_act in tower_plan(params, prefix):
        x = F.linear(x, w, b)
        if has_act:
            x = act(x)
    return x


# ----------------------------------------------------------------------------- message
def edge_matrices(params, bfm, mf, nf, act=F.relu):
    """A(e_ij) for every pair, (B,N,N,mf,nf) -- the legacy layout of edge_network.py:40."""
    B, N = bfm.shape[:2]
    return edge_map(params, bfm.reshape(-1, bfm.shape[-1]), act).view(B, N, N, mf, nf)


def edge_network_pair(params, afm, bfm, act=F.relu):
    """Per-pair messages m_ij = A(e_ij) h_j, (B,N,N,mf); the contract of edge_network.py:40,52
    that every aggregator documents (adjacent_message_agg.py:13).  No bias in this form."""
    mf = params["message_bias"].shape[0]
    nf = afm.shape[-1]
    A = edge_matrices(params, bfm, mf, nf, act)
    return torch.einsum("bijmn,bjn->bijm", A, afm)


def edge_network_fused(params, afm, bfm, act=F.relu):
    """HEAD behaviour (edge_network.py:30-51): m_i = sum over ALL j (non-edges and padding
    included) of A(e_ij) h_j, plus message_bias.  Written as the same block matvec the
    reference performs so the summation structure is the same."""
    mf = params["message_bias"].shape[0]
    B, N, nf = afm.shape
    A = edge_matrices(params, bfm, mf, nf, act)
    blk = A.permute(0, 1, 3, 2, 4).reshape(B, N * mf, N * nf)
    m = torch.bmm(blk, afm.reshape(B, N * nf, 1)).view(B, N, mf)
    return m + params["message_bias"]


def att_edge_network_pair(params, afm, bfm, act=F.relu, attn_act=None):
    """AttEdgeNetwork (att_edge_network.py:13-31): gate_ij = attn_act(Linear([h_i, e_ij])) over
    the FEATURE axis, x_ij = gate_ij * h_j, m_ij = A(e_ij) x_ij.  (B,N,N,mf)."""
    mf = params["message_bias"].shape[0]
    B, N, nf = afm.shape
    A = edge_matrices(params, bfm, mf, nf, act)
    hi = afm.unsqueeze(2).expand(B, N, N, nf)              # h_i repeated along j
    z = F.linear(torch.cat([hi, bfm], dim=-1), params["attn.weight"], params["attn.bias"])
    gate = torch.softmax(z, dim=-1) if attn_act is None else attn_act(z)
    x = gate * afm.unsqueeze(1)                            # h_j along axis 2
    return torch.einsum("bijmn,bijn->bijm", A, x)


def ggnn_fused(params, afm, ibfm):
    """GGNNMsgPass (ggnn_msg_pass.py:17-31): integer bond type indexes [zeros; adj_w]; fused
    all-pairs sum plus bias, like edge_network_fused."""
    table = torch.cat([torch.zeros_like(params["adj_w"][:1]), params["adj_w"]], dim=0)
    A = table[ibfm]                                         # (B,N,N,mf,nf)
    return torch.einsum("bijmn,bjn->bim", A, afm) + params["message_bias"]


def bilinear_pair(afm, bfm):
    """BiLiniearEdgeNetwork (bilinear_edge_network.py:25-37): bfm_ij viewed (nf, nf*nf);
    v = h_j^T T -> (nf,nf); m_ij = v h_i.  Output is .squeeze()d like the reference."""
    B, N, nf = afm.shape
    T = bfm.view(B, N, N, nf, nf, nf)
    # reference: afm.unsqueeze(1) broadcasts over axis 1 => indexed by axis-2 atom (j);
    #            afm.unsqueeze(2) broadcasts over axis 2 => indexed by axis-1 atom (i)
    v = torch.einsum("bja,bijakc->bijkc", afm, T)
    return torch.einsum("bijkc,bic->bijk", v, afm).squeeze()


# ----------------------------------------------------------------------------- aggregators
def agg_adj(messages, adj):
    """AdjMsgAgg (adjacent_message_agg.py:18): sum_j adj_ij m_ij."""
    return (messages * adj.unsqueeze(-1)).sum(dim=-2)


def agg_wadj(messages, adj):
    """WAdjMsgAgg (weighted_adjacent_message_agg.py:20): softmax over the padded row of adj."""
    return (messages * torch.softmax(adj, dim=-1).unsqueeze(-1)).sum(dim=-2)


def agg_att(params, messages, adj, attn_act=None):
    """AttMsgAgg (attention_message_agg.py:9-24): weight_ij = act(w*adj_ij + b); the default act
    is a softmax over a size-1 axis, i.e. weight == 1 for every pair."""
    z = F.linear(adj.unsqueeze(-1), params["att.0.weight"], params["att.0.bias"])
    w = torch.softmax(z, dim=-1) if attn_act is None else attn_act(z)
    return (messages * w).sum(dim=-2)


# ----------------------------------------------------------------------------- update
def gru_cell(params, m, h, mask, prefix="gru_cell."):
    """GRUCell.forward (gru_update.py:26-35); weights stored (in, 3H), gate order r,z,n."""
    gi = m @ params[prefix + "weight_ih"] + params[prefix + "bias_ih"]
    gh = h @ params[prefix + "weight_hh"] + params[prefix + "bias_hh"]
    H = h.shape[-1]
    ri, zi, ni = gi.split(H, dim=-1)
    rh, zh, nh = gh.split(H, dim=-1)
    r = torch.sigmoid(ri + rh) * mask
    z = torch.sigmoid(zi + zh) * mask
    n = torch.tanh(ni + r * nh) * mask
    return (1 - z) * n + z * h


def gru_update(params, messages, node_states, mask):
    """GRUUpdate.forward (gru_update.py:55-68): flatten, cell, multiply by mask again."""
    H = node_states.shape[-1]
    mk = mask.reshape(-1, 1)
    out = gru_cell(params, messages.reshape(-1, messages.shape[-1]), node_states.reshape(-1, H), mk)
    return (out * mk).view(node_states.shape)


# ----------------------------------------------------------------------------- norms / readout
def mask_bn1d(x, mask, weight=None, bias=None, running_mean=None, running_var=None,
              training=True, momentum=0.1, eps=1e-5):
    """MaskBatchNorm1d.forward (models/mask_batch_norm.py:20-38).  Returns (y, new_rm, new_rv).
    Masked mean, biased masked variance, eps added OUTSIDE the square root."""
    mk = mask.reshape(-1, 1)
    y = x.reshape(-1, x.shape[-1])
    cnt = mk.sum()
    mean = (y * mk).sum(dim=0) / cnt
    var = (((y - mean) * mk) ** 2).sum(dim=0) / cnt
    new_rm, new_rv = running_mean, running_var
    if not training and running_mean is not None:
        y = (y - running_mean) / (running_var ** 0.5 + eps)
    else:
        if running_mean is not None:
            new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
            new_rv = (1 - momentum) * running_var + momentum * var.detach()
        y = (y - mean) / (var.sqrt() + eps)
    if weight is not None:
        y = weight * y + bias
    return (y * mk).view(x.shape), new_rm, new_rv


def mask_bn(x, mask, eps=1e-6):
    """MaskBatchNorm.forward (models/mask_batch_norm.py:9-15): the mean numerator is NOT masked."""
    mk = mask.reshape(-1, 1)
    y = x.reshape(-1, x.shape[-1])
    cnt = mk.sum()
    mean = y.sum(dim=0) / cnt
    c = (y - mean) * mk
    var = (c ** 2).sum(dim=0) / cnt
    return (c / (var + eps).sqrt()).view(x.shape)


def graph_level_output(params, x, mask=None):
    """GraphLevelOutput.forward (mpnn_functions/readout/graph_level_output.py:30-47)."""
    def lin(name, v):
        return F.linear(v, params[name + ".0.weight"], params[name + ".0.bias"])
    if mask is not None:
        xm = x * mask
        g = torch.softmax(lin("i", xm), dim=-1) * lin("j", xm) * mask
    else:
        g = torch.softmax(lin("i", x).sum(dim=1), dim=-1).unsqueeze(1) * lin("j", x)
    return g.sum(dim=1)


def set2vec(params, x, mask=None, steps=100, inner_prod="default"):
    """Set2Vec.forward (mpnn_functions/readout/set2vec.py:93-151) with LSTMCellHidden.forward (:67-75).
    PARITY UNPINNED for this function: the reference module cannot be imported here (it imports
    pre_process.utils -> rdkit, and uses Py2 dict.iteritems), so this restatement follows the source text only.
    x (B,N,nf2); params: q_attn.weight, e_attn.weight, lstmcell.{w,b}_h{i,f,g,o}.  Returns (B, 2*nf2)."""
    B, nf = x.shape[0], x.shape[-1]
    mprev = torch.cat([x.new_zeros(B, nf), x.new_zeros(B, nf)], dim=1)          # :111-113
    cprev = x.new_zeros(B, nf)
    pen = (1 - mask) * _BIG_NEGATIVE if mask is not None else None              # :121-123
    m = mprev
    for _ in range(steps):
        def gate(g, fn):
            return fn(mprev.matmul(params["lstmcell.w_h" + g]) + params["lstmcell.b_h" + g])
        i, f, g, o = gate("i", torch.sigmoid), gate("f", torch.sigmoid), gate("g", torch.tanh), gate("o", torch.sigmoid)
        c = f * cprev + i * g
        m = o * torch.tanh(c)
        query = F.linear(m, params["q_attn.weight"]).unsqueeze(1)               # :130
        if inner_prod == "default":
            energies = F.linear(torch.tanh(query + x).view(-1, nf), params["e_attn.weight"])   # :133
        else:
            energies = x.matmul(query.view(-1, nf, 1)).view(B, -1)              # :136
        if pen is not None:
            energies = energies + pen.view(-1, 1)
        att = torch.softmax(energies, dim=0).view(B, -1, 1)                     # :139 (dim 0 of the flat batch)
        read = att.mul(x).sum(dim=1)
        m = torch.cat([m, read], dim=1)
        mprev, cprev = m, c
    return m


# ----------------------------------------------------------------------------- models
def basic_model_forward(params, afm, bfm, adj, mask, steps=3, return_state=False):
    """The intended basic_model.BasicModel.forward (models/basic_model.py:50-58):
    node_state = uf(ma(mf(afm, bfm), adj), node_state, mask), T times, message always from the
    ORIGINAL afm; readout on cat[node_state, afm].  `params` = graph_model.* entries."""
    mfp, ufp, ofp = sub(params, "mf."), sub(params, "uf."), sub(params, "of.")
    pair = edge_network_pair(mfp, afm, bfm)        # edge_embed is cached across steps (:57)
    h = afm
    for _ in range(steps):
        h = gru_update(ufp, agg_adj(pair, adj), h, mask)
    out = graph_level_output(ofp, torch.cat([h, afm], dim=-1), mask)
    return (out, h) if return_state else out


def lipo_model_forward(params, batch, steps=6, training=True, return_buffers=False):
    """graph_norm_wrapper.GraphWrapper + lipo_basic_model.BasicModel
    (models/graph_norm_wrapper.py:12-13, models/lipo_basic_model.py:81-86).
    `params` is the wrapper's state_dict (keys bn.*, graph_model.*).  BatchNorm running
    statistics are threaded functionally and returned when `return_buffers`."""
    buf = {k: v for k, v in params.items() if "running_" in k}

    def bn(prefix, x, mask):
        y, rm, rv = mask_bn1d(x, mask, params[prefix + "weight"], params[prefix + "bias"],
                              buf[prefix + "running_mean"], buf[prefix + "running_var"], training)
        buf[prefix + "running_mean"], buf[prefix + "running_var"] = rm, rv
        return y

    mask = batch["mask"]
    afm = torch.cat([batch["afm"], bn("bn.", batch["nafm"], mask)], dim=-1)
    gm = "graph_model."
    mfp, ufp, ofp = sub(params, gm + "mf."), sub(params, gm + "uf."), sub(params, gm + "of.")
    msg = edge_network_fused(mfp, afm, batch["bfm"])
    h = afm
    for _ in range(steps):
        h = bn(gm + "bn.", gru_update(ufp, bn(gm + "ma_bn.", msg, mask), h, mask), mask)
    out = graph_level_output(ofp, torch.cat([h, afm], dim=-1), mask)
    return (out, buf) if return_buffers else out


def att_model_forward(params, afm, bfm, adj, mask, steps, return_state=False):
    """models/att_model.py:55-59: one AttEdgeNetwork PER STEP (modules mf0..mf{T-1}), AdjMsgAgg, GRU,
    parameter-free MaskBatchNorm; message always from the original afm; readout on cat[state, afm]
    (GraphLevelOutput here; the reference's default Set2Vec is outside the hot path)."""
    ufp, ofp = sub(params, "uf."), sub(params, "of.")
    h = afm
    for i in range(steps):
        pair = att_edge_network_pair(sub(params, "mf%d." % i), afm, bfm)
        h = mask_bn(gru_update(ufp, agg_adj(pair, adj), h, mask), mask)
    out = graph_level_output(ofp, torch.cat([h, afm], dim=-1), mask)
    return (out, h) if return_state else out


# ----------------------------------------------------------------------------- index oracle
def dense_to_csr(adj):
    """Bit-exact index oracle: rows = b*N+i, columns = b*N+j, in adj.nonzero() order."""
    B, N = adj.shape[:2]
    nz = adj.nonzero()
    rows = nz[:, 0] * N + nz[:, 1]
    counts = torch.bincount(rows, minlength=B * N)
    row_ptr = torch.zeros(B * N + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    col_idx = (nz[:, 0] * N + nz[:, 2]).to(torch.int32)
    return row_ptr, col_idx, adj[nz[:, 0], nz[:, 1], nz[:, 2]]
