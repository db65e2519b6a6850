"""Parity cases added in round 3:

* the configuration-3 model (models/att_model.py:55-59) against vectors recorded from the REAL reference modules
  (tests/golden/model_att_*.npz), forward and every gradient;
* GraphLevelOutput without a mask (graph_level_output.py:38-47) against its reference vector;
* configuration 5's defining feature -- hub atoms of preferential-attachment molecules of ~64 atoms at hidden 256 --
  against the dense oracle (the dense path holds N^2 * H^2 floats per molecule: 1 GB at N = 64);
* whole training steps at the FULL sizes of configs[4] (hidden 256, 50 k skewed molecules) and configs[2] (attention
  model, hidden 128, 5 steps, 100 k molecules) against float64 restatements on the device.
"""
import numpy as np
import pytest
import torch

from conftest import Fixture, max_err, record_parity
from oracle import dense_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    return max_err(a, b) / max(1.0, float(torch.as_tensor(b).detach().abs().max()))


# ------------------------------------------------------------------------------------------- reference fixtures
@pytest.mark.parametrize("tag,H,ef,T", [("h8_T3", 8, 4, 3), ("h22_T5", 22, 7, 5)])
def test_att_model_against_reference_fixture(dev, tag, H, ef, T):
    from mpnn_amd.models.att_model import BasicModel as AttModel
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    f = Fixture("model_att_" + tag)
    model = AttModel(H, ef, H, 9, 6, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T,
                     readout_func=GraphLevelOutput).to(dev)
    model.load_state_dict(f.params)
    model.train()
    i = {k: v.to(dev) for k, v in f.inputs.items()}
    i["afm"].requires_grad_(True)
    out = model(i["afm"], i["bfm"], i["adj"], i["mask"])
    state, _ = model.message_passing(i["afm"], i["bfm"], i["adj"], i["mask"])
    (out * f.cot.to(dev)).sum().backward()
    e_state, e_out = max_err(state.detach().cpu(), f.out["node_state"]), max_err(out.detach().cpu(), f.out[""])
    record_parity("att_model_fixture_" + tag, node_state_max_abs_err=e_state, readout_max_abs_err=e_out,
                  node_state_max_abs=float(f.out["node_state"].abs().max()), bar=1e-5)
    assert e_state < 1e-5 and e_out < 1e-5      # north_star's bar; measured 9.5e-7 / 1.2e-7 (profiles/r04_parity.json)
    assert _rel(i["afm"].grad.cpu(), f.gin["afm"]) < 1e-4
    seen = 0
    for k, p in model.named_parameters():
        if k in f.gp:
            assert p.grad is not None, k
            assert _rel(p.grad.cpu(), f.gp[k]) < 2e-4, (k, _rel(p.grad.cpu(), f.gp[k]))
            seen += 1
    assert seen >= 4 * T + 4


def test_graph_level_output_without_mask(dev):
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    f = Fixture("graph_level_output_nomask")
    m = GraphLevelOutput(8, 6).to(dev)
    m.load_state_dict(f.params)
    x = f.inputs["x"].to(dev).requires_grad_(True)
    out = m(x)
    (out * f.cot.to(dev)).sum().backward()
    assert max_err(out.detach().cpu(), f.out[""]) < 1e-5
    assert _rel(x.grad.cpu(), f.gin["x"]) < 1e-5
    for k, p in m.named_parameters():
        if k in f.gp:
            assert _rel(p.grad.cpu(), f.gp[k]) < 2e-5, k


# ------------------------------------------------------------------------------------------- config 5: hubs
def test_c5_hub_molecules_against_oracle(dev):
    """Two preferential-attachment molecules of 56-64 atoms (the largest the dense oracle takes at hidden 256: 1 GB of
    edge matrices per molecule), each with a hub of degree >= 12: forward, final state and every gradient."""
    from mpnn_amd import synth
    from mpnn_amd.models.basic_model import BasicModel
    from mpnn_amd.models.graph_model_wrapper import GraphWrapper
    H, T = 256, 3
    mb = synth.make_molecules(400, H, seed=317 + H, dist="skewed")
    deg = np.diff(mb.row_ptr)
    hub = np.array([deg[mb.atom_ptr[g]:mb.atom_ptr[g + 1]].max() for g in range(mb.num_mols)])
    ok = np.nonzero((mb.n_atoms >= 56) & (mb.n_atoms <= 64) & (hub >= 12))[0]
    assert len(ok) >= 2, "generator no longer produces 56-64-atom molecules with a degree-12 hub"
    sub = synth.select(mb, ok[:2])
    assert int(np.diff(sub.row_ptr).max()) >= 12
    batch = {k: torch.from_numpy(v) for k, v in synth.to_dense(sub).items()}
    N = batch["adj"].shape[-1]
    torch.manual_seed(317)
    model = GraphWrapper(BasicModel(H, 4, H, N, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                                    message_steps=T))
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "bias" in k:
                p.uniform_(-0.1, 0.1)
    leaves, params = {}, {}
    for k, v in model.state_dict(keep_vars=True).items():
        if id(v) not in leaves:
            leaves[id(v)] = v.detach().cpu().clone().requires_grad_(v.requires_grad and v.is_floating_point())
        params[k] = leaves[id(v)]
    cot = torch.rand(2, 8) - 0.5
    ref, ref_state = O.basic_model_forward(O.sub(params, "graph_model."), batch["afm"], batch["bfm"], batch["adj"],
                                           batch["mask"], T, True)
    (ref * cot).sum().backward()
    model = model.to(dev)
    gb = {k: v.to(dev) for k, v in batch.items()}
    out = model(gb)
    state, _ = model.graph_model.message_passing(gb["afm"], gb["bfm"], gb["adj"], gb["mask"])
    (out * cot.to(dev)).sum().backward()
    assert _rel(state.detach().cpu(), ref_state) < 1e-5
    assert _rel(out.detach().cpu(), ref) < 2e-5
    checked = 0
    for k, p in model.named_parameters():
        g_ref = params[k].grad
        if g_ref is None or p.grad is None:
            continue
        assert _rel(p.grad.cpu(), g_ref) < 2e-4, (k, _rel(p.grad.cpu(), g_ref))
        checked += 1
    assert checked >= 8


# ------------------------------------------------------------------------------------------- float64 on the device
def _gru64(m, h, mask, W_ih, W_hh, b_ih, b_hh):
    H = h.shape[1]
    gi = m @ W_ih + b_ih
    gh = h @ W_hh + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mask
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mask
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mask
    return ((1 - z) * n + z * h) * mask


def _tower64(mf, p64, prefix, rows):
    """edge_map of module `mf` on `rows` in float64 with the parameters p64[prefix + ...]; the 50 aliased layers
    (edge_network.py:20) read the ONE tensor stored under the first alias' name."""
    x = rows
    mods = list(mf.edge_map)
    for i, mod in enumerate(mods):
        if isinstance(mod, torch.nn.Linear):
            x = x @ p64[prefix + "edge_map.%d.weight" % i].t() + p64[prefix + "edge_map.%d.bias" % i]
        elif isinstance(mod, torch.nn.Sequential):
            first = next(j for j, mm in enumerate(mods) if mm is mod)
            x = torch.relu(x @ p64[prefix + "edge_map.%d.0.weight" % first].t())
        else:
            x = torch.relu(x)
    return x


def _condition(model, mfs, g):
    """Weights that carry signal: kaiming towers (as test_lipo.py:132 initialises them), non-zero biases everywhere,
    last tower layer scaled so the bond matrices are O(1/4)."""
    with torch.no_grad():
        for mf in mfs:
            for mod in mf.edge_map.modules():
                if isinstance(mod, torch.nn.Linear):
                    torch.nn.init.kaiming_uniform_(mod.weight, nonlinearity="relu")
        for n, p in model.named_parameters():
            if n.endswith("bias") or "bias_" in n:
                p.uniform_(-0.05, 0.05)
        for mf in mfs:
            A_now, _ = mf._edge_matrices(g)
            last = mf.edge_map[-1]
            sc = 0.25 / float(A_now.abs().max())
            last.weight.mul_(sc)
            last.bias.mul_(sc)


def _compare_param_grads(model, p64, skip=("of.",), tol=5e-4):
    seen, checked = set(), 0
    for n, p in model.named_parameters():
        if n.startswith(skip) or p.data_ptr() in seen or n.endswith("message_bias"):
            continue
        seen.add(p.data_ptr())
        want = p64[n].grad
        if want is None:
            continue
        scale = max(1e-3, float(want.abs().max()))
        assert max_err(p.grad, want) / scale < tol, (n, max_err(p.grad, want) / scale)
        checked += 1
    return checked


def test_c5_basic_model_training_step_against_float64(dev):
    """configs[4] at full size: 50 k preferential-attachment molecules of 10-200 atoms, hidden 256, 3 rounds of
    message -> sum -> GRU and the backward pass from a random cotangent.  The float64 restatement walks the batch in
    chunks of 2,500 molecules (molecules are independent; parameter gradients add up over chunks)."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel
    H, T, G = 256, 3, 50_000
    mb = synth.make_molecules(G, H, seed=317, dist="skewed")
    g = MolGraph.from_molbatch(mb, dev)
    h = torch.from_numpy(mb.atom_feat).to(dev)
    V = g.num_nodes
    assert int((g.row_ptr[1:] - g.row_ptr[:-1]).max()) > 40
    torch.manual_seed(5)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                       message_steps=T).to(dev)
    _condition(model, [model.mf], g)
    mask = torch.ones(V, 1, device=dev)
    cot = torch.randn(V, H, device=dev) / V ** 0.5
    state, _ = model.message_passing(h, g, g, mask)
    state.backward(cot)
    state = state.detach()

    p64 = {n: p.detach().double().requires_grad_(True) for n, p in model.named_parameters() if not n.startswith("of.")}
    atom_ptr = torch.from_numpy(mb.atom_ptr).to(dev)
    row_ptr = g.row_ptr.long()
    src_all, typ_all = g.col_idx.long(), g.edge_type.long()
    rows = torch.cat([g.type_feat.new_zeros(1, 4), g.type_feat]).double()
    worst = 0.0
    step = 2_500
    for m0 in range(0, G, step):
        a0, a1 = int(atom_ptr[m0]), int(atom_ptr[min(m0 + step, G)])
        e0, e1 = int(row_ptr[a0]), int(row_ptr[a1])
        A = _tower64(model.mf, p64, "mf.", rows).view(-1, H, H)[1:]
        src = src_all[e0:e1] - a0
        dst = torch.repeat_interleave(torch.arange(a1 - a0, device=dev), (row_ptr[a0 + 1:a1 + 1] - row_ptr[a0:a1]))
        typ = typ_all[e0:e1]
        h64 = h[a0:a1].double()
        agg = torch.zeros(a1 - a0, H, dtype=torch.float64, device=dev)
        for k in range(A.shape[0]):
            idx = (typ == k).nonzero().squeeze(1)
            agg = agg.index_add(0, dst[idx], h64[src[idx]] @ A[k].t())
        st = h64
        for _ in range(T):
            st = _gru64(agg, st, mask[a0:a1].double(), p64["uf.gru_cell.weight_ih"], p64["uf.gru_cell.weight_hh"],
                        p64["uf.gru_cell.bias_ih"], p64["uf.gru_cell.bias_hh"])
        st.backward(cot[a0:a1].double())
        worst = max(worst, max_err(state[a0:a1], st.detach()))
    record_parity("c5_full_size_training_step", node_state_max_abs_err=worst, node_state_max_abs=float(state.abs().max()),
                  atoms=V, hidden=H, steps=T, bar=1e-5)
    assert worst < 1e-5, worst                                # measured 4.8e-6 (profiles/r04_parity.json)
    assert _compare_param_grads(model, p64) >= 6


def test_c3_attention_model_training_step_against_float64(dev):
    """configs[2] at full size: the attention model (an AttEdgeNetwork per step, AdjMsgAgg, GRU, the parameter-free
    masked norm; models/att_model.py:55-59) at hidden 128, 5 steps, 100 k molecules: final node state and the gradient
    of every hot-path parameter against a float64 restatement on the device.  The norm couples every atom of the batch,
    so the restatement runs the whole batch, one checkpointed segment per step."""
    from torch.utils.checkpoint import checkpoint
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.att_model import BasicModel as AttModel
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    H, T, G = 128, 5, 100_000
    mb = synth.make_molecules(G, H, seed=317)
    g = MolGraph.from_molbatch(mb, dev)
    h = torch.from_numpy(mb.atom_feat).to(dev)
    V = g.num_nodes
    torch.manual_seed(6)
    model = AttModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T,
                     readout_func=GraphLevelOutput).to(dev)
    _condition(model, model.mfs, g)
    mask = torch.ones(V, 1, device=dev)
    cot = torch.randn(V, H, device=dev) / V ** 0.5
    state, _ = model.message_passing(h, g, g, mask)
    state.backward(cot)
    state = state.detach()

    p64 = {n: p.detach().double().requires_grad_(True) for n, p in model.named_parameters() if not n.startswith("of.")}
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    by_type = [(typ == k).nonzero().squeeze(1) for k in range(g.num_types)]
    rows = torch.cat([g.type_feat.new_zeros(1, 4), g.type_feat]).double()
    tf64, h64, mask64 = g.type_feat.double(), h.double(), mask.double()

    def raw_step(st, i):
        pre = "mf%d." % i
        A = _tower64(model.mfs[i], p64, pre, rows).view(-1, H, H)[1:]
        Wa, ba = p64[pre + "attn.weight"], p64[pre + "attn.bias"]
        z_atom = h64 @ Wa[:, :H].t() + ba                    # destination-atom part of Linear([h_i, e_ij])
        q = tf64 @ Wa[:, H:].t()                             # bond part, one row per bond type
        agg = torch.zeros(V, H, dtype=torch.float64, device=dev)
        for k, idx in enumerate(by_type):
            gate = torch.softmax(z_atom[dst[idx]] + q[k], dim=-1)
            agg = agg.index_add(0, dst[idx], (gate * h64[src[idx]]) @ A[k].t())
        return agg, _gru64(agg, st, mask64, p64["uf.gru_cell.weight_ih"], p64["uf.gru_cell.weight_hh"],
                           p64["uf.gru_cell.bias_ih"], p64["uf.gru_cell.bias_hh"])

    def norm64(y):
        mean = y.sum(0) / mask64.sum()                       # MaskBatchNorm: unmasked numerator (mask_batch_norm.py:13)
        c = (y - mean) * mask64
        var = (c ** 2).sum(0) / mask64.sum()
        return c / (var + 1e-6).sqrt(), var

    def one_step(st, i, *_params):
        return norm64(raw_step(st, i)[1])[0]

    plist = list(p64.values())
    st = h64.clone().requires_grad_(True)                    # a leaf that requires grad: checkpoint needs one
    cur = st
    for i in range(T):
        cur = checkpoint(one_step, cur, i, *plist, use_reentrant=False)
    cur.backward(cot.double())
    e_final = max_err(state, cur.detach())
    # Where the error comes from.  (1) the operators themselves: step 0's message + sum and raw update output on the HIP
    # path against float64 on the same inputs; (2) the norm divides by sqrt(var): whatever error the update leaves is
    # multiplied by 1 / std of its column, T times over; (3) the same model with the standalone norm kernels.
    with torch.no_grad():
        agg64, y64 = raw_step(h64, 0)
        mf0 = model.mfs[0]
        mf0.bind_graph(g)
        agg = model.ma(mf0(h, g), g).reshape(V, H)
        cell = model.uf.gru_cell
        from mpnn_amd import ops
        y = ops.gru_update(agg, h, mask.reshape(-1), cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
        e_agg, e_y = max_err(agg, agg64), max_err(y, y64)
        n64, var0 = norm64(y64)
        st64 = h64
        amp = 0.0
        for i in range(T):
            st64, var_i = norm64(raw_step(st64, i)[1])
            amp = max(amp, float((var_i + 1e-6).rsqrt().max()))
        e_rounded = max_err(norm64(y64.float().double())[0], n64)      # a float32-ROUNDED exact y through the exact norm
        model.fuse_norm = False
        st_unfused, _ = model.message_passing(h, g, g, mask)
        model.fuse_norm = True
        e_unfused = max_err(st_unfused, cur.detach())
    record_parity("c3_full_size_training_step", node_state_max_abs_err=e_final, node_state_max_abs=float(state.abs().max()),
                  standalone_norm_kernels_max_abs_err=e_unfused, step0_message_sum_max_abs_err=e_agg,
                  step0_message_sum_max_abs=float(agg64.abs().max()), step0_raw_update_max_abs_err=e_y,
                  largest_one_over_std_of_a_normalised_column=amp,
                  exact_update_rounded_to_float32_then_exact_norm_max_abs_err=e_rounded, atoms=V, hidden=H, steps=T, bar=1e-5)
    assert e_final < 1e-5                                     # measured 5.6e-6 on states of up to 12.6 (profiles/r04_parity.json)
    assert _compare_param_grads(model, p64, tol=1e-3) >= 4 * T + 4


# ------------------------------------------------------------------------------------------- sparse collate
def test_lipo_model_gives_equal_outputs_behind_both_collates(dev):
    """The same 16 per-molecule objects through the dense contract of the reference's collate
    (pre_process/data_loader.py:50-70) and through mpnn_amd.collate.collate_sparse: the test_lipo.py model
    (graph_norm wrapper + lipo BasicModel, training-mode batch norms) must give the same outputs and gradients, and
    the dense route must match the oracle."""
    from test_collate_cpu import dense_contract, fake_graphs
    from mpnn_amd.collate import collate_2d_graphs, collate_sparse
    from mpnn_amd.models.graph_norm_wrapper import GraphWrapper
    from mpnn_amd.models.lipo_basic_model import BasicModel
    af, naf, ef, T = 19, 3, 7, 3
    graphs = fake_graphs(16, seed=11, af=af, naf=naf, ef=ef)
    torch.manual_seed(317)
    model = GraphWrapper(BasicModel(af + naf, ef, af + naf, 50, 2 * af, message_opts={}, agg_opts={}, update_opts={},
                                    readout_opts={}, message_steps=T), naf)
    model.apply(BasicModel.init_weights)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "bias" in k:
                p.uniform_(-0.05, 0.05)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dense_np = dense_contract(graphs)
    ref = O.lipo_model_forward(params, {k: torch.from_numpy(v) for k, v in dense_np.items() if k != "labels"}, steps=T,
                               training=True)
    model = model.to(dev).train()
    cot = (torch.rand(16, 2 * af) - 0.5).to(dev)

    def run(batch):
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.zero_grad()
        out = model(batch)
        (out * cot).sum().backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        model.load_state_dict(sd)                            # undo the running-statistics update
        return out.detach(), grads

    dense_dev = {k: torch.from_numpy(v).to(dev) for k, v in dense_np.items()}
    adapter = collate_2d_graphs(graphs, dev)                 # the adapter = the dense contract, built on the device
    for k, v in dense_dev.items():
        assert torch.equal(adapter[k], v.to(adapter[k].dtype)), k
    out_d, g_d = run(dense_dev)
    out_a, _ = run(adapter)
    out_s, g_s = run(collate_sparse(graphs, dev))
    record_parity("lipo_model_behind_both_collates", dense_vs_oracle_max_abs_err=max_err(out_d.cpu(), ref),
                  sparse_vs_dense_max_abs_err=max_err(out_s, out_d), adapter_vs_dense_max_abs_err=max_err(out_a, out_d),
                  readout_max_abs=float(ref.abs().max()), bar=1e-5)
    assert max_err(out_a, out_d) < 2e-6                      # same tensors in: equal up to the norms' atomic sum order
    assert max_err(out_d.cpu(), ref) < 1e-5                  # measured 2.1e-6 on readouts of up to 5 (profiles/r04_parity.json)
    assert max_err(out_s, out_d) < 1e-5
    for k in g_d:
        assert _rel(g_s[k], g_d[k]) < 2e-4, k
