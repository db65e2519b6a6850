"""BASELINE.json-sized runs checked through size-independent properties (the oracle cannot run at
these sizes): checksums in fp64, linearity, exact closed forms, permutation equivariance, and the
dense->CSR kernel against the generator's own CSR."""
import numpy as np
import pytest
import torch

from conftest import max_err, record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def c2(dev):
    """configs[1]: 100k molecules, ~30 atoms / 60 edges, hidden 64."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(100_000, 64, seed=317)
    return mb, MolGraph.from_molbatch(mb, dev), torch.from_numpy(mb.atom_feat).to(dev)


def test_c2_aggregator_checksum_and_linearity(dev, c2):
    from mpnn_amd import ops
    mb, g, h = c2
    E, V = g.num_edges, g.num_nodes
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(E, 64, device=dev, generator=gen)
    y = torch.randn(E, 64, device=dev, generator=gen)
    w = torch.rand(E, device=dev, generator=gen) + 0.5
    ox = ops.segsum_raw(x, g.row_ptr, w, V)
    oy = ops.segsum_raw(y, g.row_ptr, w, V)
    oxy = ops.segsum_raw(2.0 * x - 3.0 * y, g.row_ptr, w, V)
    assert max_err(oxy, 2.0 * ox - 3.0 * oy) < 1e-4                      # linearity (rows have <= 6 terms)
    # checksum of checksums: column sums of the output == weighted column sums of the input
    assert max_err(ox.double().sum(0), (x.double() * w.double().unsqueeze(1)).sum(0)) < 1e-6 * E
    # exact identity on integer data: plain sum of ones == degree
    deg = ops.segsum_raw(torch.ones(E, 4, device=dev), g.row_ptr, None, V)
    assert torch.equal(deg[:, 0].to(torch.int32), g.row_ptr[1:] - g.row_ptr[:-1])


def test_c2_edge_message_checksum(dev, c2):
    from mpnn_amd import ops
    mb, g, h = c2
    gen = torch.Generator(device=dev).manual_seed(2)
    A = torch.randn(g.num_types, 64, 64, device=dev, generator=gen) / 8.0
    msg = ops.edge_message_raw(h, A, g)
    # sum_e msg[e] == sum_k A_k (sum_{e of type k} h[src e])   in fp64
    src = g.col_idx.long()
    S = torch.zeros(g.num_types, 64, dtype=torch.float64, device=dev).index_add_(0, g.edge_type.long(), h[src].double())
    ref = torch.einsum("kmn,kn->m", A.double(), S)
    assert max_err(msg.double().sum(0), ref) < 1e-6 * g.num_edges
    # spot rows against a direct fp64 product
    pick = torch.randint(0, g.num_edges, (4096,), device=dev, generator=gen)
    direct = torch.einsum("emn,en->em", A[g.edge_type.long()[pick]].double(), h[src[pick]].double())
    assert max_err(msg[pick], direct) < 1e-5


def test_c2_gru_closed_forms(dev, c2):
    from mpnn_amd import ops
    mb, g, h = c2
    V, H = h.shape
    z3 = torch.zeros(H, 3 * H, device=dev)
    b0 = torch.zeros(3 * H, device=dev)
    mask = (torch.arange(V, device=dev) % 7 != 0).float()
    m = torch.randn(V, H, device=dev)
    out, _ = ops.gru_update_raw(m, h, mask, z3, z3, b0, b0, False)
    # zero weights: r = z = 1/2, n = tanh(0) = 0  =>  out = h/2 on real atoms, 0 on masked ones
    assert max_err(out, 0.5 * h * mask.unsqueeze(1)) < 1e-6
    # update gate forced shut (z -> 1): the state passes through unchanged
    big = b0.clone()
    big[H:2 * H] = 40.0
    out, _ = ops.gru_update_raw(m, h, None, z3, z3, big, b0, False)
    assert max_err(out, h) < 1e-6


def test_c2_model_is_permutation_equivariant_over_molecules(dev):
    """Shuffling the molecules of a batch permutes the per-molecule outputs and nothing else."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel
    from mpnn_amd.models.graph_model_wrapper import GraphWrapper
    mb = synth.make_molecules(20_000, 64, seed=9)
    perm = np.random.default_rng(0).permutation(mb.num_mols)
    mb2 = synth.select(mb, perm)
    torch.manual_seed(1)
    model = GraphWrapper(BasicModel(64, 4, 64, 50, 8, message_opts={}, agg_opts={}, update_opts={},
                                    readout_opts={})).to(dev)
    outs = []
    for b in (mb, mb2):
        batch = {"afm": torch.from_numpy(b.atom_feat).to(dev), "graph": MolGraph.from_molbatch(b, dev),
                 "mask": torch.ones(b.num_atoms, 1, device=dev)}
        with torch.no_grad():
            outs.append(model(batch))
    assert max_err(outs[1], outs[0][torch.from_numpy(perm).to(dev)]) < 2e-5


def test_dense_to_csr_at_scale_equals_generator_csr(dev):
    """20k padded molecules (50 M pairs): the kernel's CSR must be the generator's CSR, bit for bit."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(20_000, 4, seed=21)
    d = synth.to_dense(mb)
    N = d["adj"].shape[1]
    g = MolGraph.from_dense(torch.from_numpy(d["adj"]).to(dev), torch.from_numpy(d["bfm"]).to(dev))
    # compact -> padded numbering
    mol = np.repeat(np.arange(mb.num_mols), mb.n_atoms)
    pad_id = mol * N + (np.arange(mb.num_atoms) - mb.atom_ptr[mol])
    deg = np.zeros(mb.num_mols * N, np.int64)
    deg[pad_id] = np.diff(mb.row_ptr)
    row_ptr = np.zeros(mb.num_mols * N + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    assert torch.equal(g.row_ptr.cpu(), torch.from_numpy(row_ptr))
    assert torch.equal(g.col_idx.cpu(), torch.from_numpy(pad_id[mb.col_idx].astype(np.int32)))
    assert g.num_types == 4
    assert torch.equal(g.type_feat[g.edge_type.long()].cpu(), torch.from_numpy(mb.type_feat[mb.bond_type]))


def test_c5_skewed_degree_aggregator(dev):
    """configs[4] shape: 10-200 atoms, preferential-attachment hubs, hidden 256."""
    from mpnn_amd import ops, synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(12_000, 8, seed=5, dist="skewed")
    g = MolGraph.from_molbatch(mb, dev)
    assert int((g.row_ptr[1:] - g.row_ptr[:-1]).max()) > 40
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(g.num_edges, 256, device=dev, generator=gen)
    out = ops.segsum_raw(x, g.row_ptr, None, g.num_nodes)
    ref = torch.zeros(g.num_nodes, 256, dtype=torch.float64, device=dev).index_add_(0, g.edge_dst.long(), x.double())
    assert max_err(out, ref) < 1e-5 * float(ref.abs().max())
    A = torch.randn(4, 256, 256, device=dev, generator=gen) / 16.0
    hh = torch.randn(g.num_nodes, 256, device=dev, generator=gen)
    msg = ops.edge_message_raw(hh, A, g)
    pick = torch.randint(0, g.num_edges, (2048,), device=dev, generator=gen)
    direct = torch.einsum("emn,en->em", A[g.edge_type.long()[pick]].double(), hh[g.col_idx.long()[pick]].double())
    assert max_err(msg[pick], direct) < 2e-5


# ------------------------------------------------------------------------------------------ hidden 128 (C3 / C4)
@pytest.fixture(scope="module")
def c4s(dev):
    """C3/C4-shaped molecules at hidden 128, 25k molecules (V ~ 750k, E ~ 1.5M): every persistent loop of the
    H = 128 kernels runs several tiles per wave and ends on a ragged tile."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(25_000, 128, seed=318)
    return mb, MolGraph.from_molbatch(mb, dev), torch.from_numpy(mb.atom_feat).to(dev)


def _gru_ref64(m, h, mask, W_ih, W_hh, b_ih, b_hh):
    H = h.shape[1]
    gi = m @ W_ih + b_ih
    gh = h @ W_hh + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mask
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mask
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mask
    return ((1 - z) * n + z * h) * mask


def test_h128_gru_forward_backward_full_size(dev, c4s):
    """GRU at hidden 128 on ~750k atoms against float64 autograd on the same device (all rows, all gradients)."""
    from mpnn_amd import ops
    mb, g, h = c4s
    V, H = h.shape
    gen = torch.Generator(device=dev).manual_seed(7)
    m = torch.randn(V, H, device=dev, generator=gen)
    mask = (torch.rand(V, 1, device=dev, generator=gen) > 0.1).float()
    bound = (6.0 / (H + 3 * H)) ** 0.5
    W_ih = (torch.rand(H, 3 * H, device=dev, generator=gen) * 2 - 1) * bound
    W_hh = (torch.rand(H, 3 * H, device=dev, generator=gen) * 2 - 1) * bound
    b_ih = torch.rand(3 * H, device=dev, generator=gen) * 0.2 - 0.1
    b_hh = torch.rand(3 * H, device=dev, generator=gen) * 0.2 - 0.1
    dout = torch.randn(V, H, device=dev, generator=gen)
    leaves = [t.clone().requires_grad_(True) for t in (m, h, W_ih, W_hh, b_ih, b_hh)]
    out = ops.gru_update(leaves[0], leaves[1], mask, *leaves[2:])
    out.backward(dout)
    ref_leaves = [t.double().requires_grad_(True) for t in (m, h, W_ih, W_hh, b_ih, b_hh)]
    ref = _gru_ref64(ref_leaves[0], ref_leaves[1], mask.double(), *ref_leaves[2:])
    ref.backward(dout.double())
    assert max_err(out.detach(), ref.detach()) < 1e-5
    for got, want, name in zip(leaves, ref_leaves, ("dm", "dh", "dW_ih", "dW_hh", "db_ih", "db_hh")):
        scale = max(1.0, float(want.grad.abs().max()))
        tol = 1e-5 if name in ("dm", "dh") else 2e-5            # weight gradients sum 750k terms
        assert max_err(got.grad, want.grad) / scale < tol, name


@pytest.mark.parametrize("gated", [False, True])
def test_h128_message_and_weight_gradient_full_size(dev, c4s, gated):
    """Typed edge message at nf = mf = 128 on ~1.5M edges: rows against float64, dx and dA against float64 autograd."""
    from mpnn_amd import ops
    mb, g, h = c4s
    E, F, K = g.num_edges, 128, int(g.type_feat.shape[0])
    gen = torch.Generator(device=dev).manual_seed(9)
    A = torch.randn(K, F, F, device=dev, generator=gen) * (1.0 / F ** 0.5)
    gate = torch.rand(E, F, device=dev, generator=gen) if gated else None
    dmsg = torch.randn(E, F, device=dev, generator=gen)
    hl, Al = h.clone().requires_grad_(True), A.clone().requires_grad_(True)
    gl = gate.clone().requires_grad_(True) if gated else None
    msg = ops.edge_message(hl, Al, g, gl)
    msg.backward(dmsg)
    src, typ = g.col_idx.long(), g.edge_type.long()
    h64, A64 = h.double().requires_grad_(True), A.double().requires_grad_(True)
    x = h64[src]
    if gated:
        g64 = gate.double().requires_grad_(True)
        x = x * g64
    ref = torch.empty(E, F, dtype=torch.float64, device=dev)
    parts = []
    for k in range(K):                                          # one dense GEMM per type
        idx = (typ == k).nonzero().squeeze(1)
        parts.append((idx, x[idx] @ A64[k].t()))
    ref = torch.zeros(E, F, dtype=torch.float64, device=dev)
    for idx, val in parts:
        ref = ref.index_add(0, idx, val)
    ref.backward(dmsg.double())
    assert max_err(msg.detach(), ref.detach()) < 1e-5
    assert max_err(hl.grad, h64.grad) / max(1.0, float(h64.grad.abs().max())) < 1e-5
    assert max_err(Al.grad, A64.grad) / max(1.0, float(A64.grad.abs().max())) < 2e-5
    if gated:
        assert max_err(gl.grad, g64.grad) / max(1.0, float(g64.grad.abs().max())) < 1e-5


@pytest.mark.parametrize("weighted", [False, True])
def test_h128_fused_message_aggregate_weight_gradient_full_size(dev, c4s, weighted):
    """dA of message + adjacency-weighted sum as one node (the no-dmsg path used in training) at hidden 128."""
    from mpnn_amd import ops
    mb, g, h = c4s
    E, V, F, K = g.num_edges, g.num_nodes, 128, int(g.type_feat.shape[0])
    gen = torch.Generator(device=dev).manual_seed(10)
    A = torch.randn(K, F, F, device=dev, generator=gen) * (1.0 / F ** 0.5)
    w = (torch.rand(E, device=dev, generator=gen) + 0.5) if weighted else None
    dagg = torch.randn(V, F, device=dev, generator=gen)
    Al = A.clone().requires_grad_(True)
    agg = ops.message_aggregate(h, Al, g, w)
    agg.backward(dagg)
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    y = dagg.double()[dst]
    if weighted:
        y = y * w.double().unsqueeze(1)
    x = h.double()[src]
    ref = torch.stack([y[typ == k].t() @ x[typ == k] for k in range(K)])
    assert max_err(Al.grad, ref) / max(1.0, float(ref.abs().max())) < 2e-5
    fwd = torch.zeros(V, F, dtype=torch.float64, device=dev)
    for k in range(K):
        idx = (typ == k).nonzero().squeeze(1)
        val = x[idx] @ A.double()[k].t()
        if weighted:
            val = val * w.double()[idx].unsqueeze(1)
        fwd = fwd.index_add(0, dst[idx], val)
    assert max_err(agg.detach(), fwd) < 2e-5


# ------------------------------------------------------------------------------------------ hidden 64 at the TIMED size
# bench.py's c2 launches put ~366 32-atom tiles on every persistent block of the H = 64 GRU backward kernel (double-
# buffered tile loop, clamped duplicate tile, one atomic flush per block) and thousands of tiles on the weight-gradient
# kernel of message+sum; the small fixtures run one tile per block.  These run the real c2 graph (V ~ 3.0 M atoms,
# E ~ 6.0 M edges) with random weights and a partial mask against float64 autograd on the device.
def test_h64_gru_forward_backward_full_size(dev, c2):
    from mpnn_amd import ops
    mb, g, h = c2
    V, H = h.shape
    assert V > 2_900_000 and H == 64
    gen = torch.Generator(device=dev).manual_seed(17)
    m = torch.randn(V, H, device=dev, generator=gen)
    mask = (torch.rand(V, 1, device=dev, generator=gen) > 0.1).float()
    bound = (6.0 / (H + 3 * H)) ** 0.5
    W_ih = (torch.rand(H, 3 * H, device=dev, generator=gen) * 2 - 1) * bound
    W_hh = (torch.rand(H, 3 * H, device=dev, generator=gen) * 2 - 1) * bound
    b_ih = torch.rand(3 * H, device=dev, generator=gen) * 0.2 - 0.1
    b_hh = torch.rand(3 * H, device=dev, generator=gen) * 0.2 - 0.1
    dout = torch.randn(V, H, device=dev, generator=gen)
    leaves = [t.clone().requires_grad_(True) for t in (m, h, W_ih, W_hh, b_ih, b_hh)]
    out = ops.gru_update(leaves[0], leaves[1], mask, *leaves[2:])
    out.backward(dout)
    with torch.no_grad():                                     # the inference instantiation (no gate dump) as well
        out_inf = ops.gru_update(m, h, mask, W_ih, W_hh, b_ih, b_hh)
    ref_leaves = [t.double().requires_grad_(True) for t in (m, h, W_ih, W_hh, b_ih, b_hh)]
    ref = _gru_ref64(ref_leaves[0], ref_leaves[1], mask.double(), *ref_leaves[2:])
    ref.backward(dout.double())
    assert max_err(out.detach(), ref.detach()) < 1e-5
    assert max_err(out_inf, ref.detach()) < 1e-5
    assert float((out.detach() * (1 - mask)).abs().max()) == 0.0          # masked atoms exactly zero
    for got, want, name in zip(leaves, ref_leaves, ("dm", "dh", "dW_ih", "dW_hh", "db_ih", "db_hh")):
        scale = max(1.0, float(want.grad.abs().max()))
        tol = 1e-5 if name in ("dm", "dh") else 2e-5            # weight gradients sum 3 M terms
        assert max_err(got.grad, want.grad) / scale < tol, name


@pytest.mark.parametrize("weighted", [False, True])
def test_h64_fused_message_aggregate_full_size(dev, c2, weighted):
    """Message + adjacency-weighted sum as one node at hidden 64 on the c2 graph: forward rows and the weight gradient
    dA (the no-dmsg path the training step takes) against float64."""
    from mpnn_amd import ops
    mb, g, h = c2
    E, V, F, K = g.num_edges, g.num_nodes, 64, int(g.type_feat.shape[0])
    gen = torch.Generator(device=dev).manual_seed(18)
    A = torch.randn(K, F, F, device=dev, generator=gen) * (1.0 / F ** 0.5)
    w = (torch.rand(E, device=dev, generator=gen) + 0.5) if weighted else None
    dagg = torch.randn(V, F, device=dev, generator=gen)
    Al = A.clone().requires_grad_(True)
    agg = ops.message_aggregate(h, Al, g, w)
    agg.backward(dagg)
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    fwd = torch.zeros(V, F, dtype=torch.float64, device=dev)
    ref = []
    for k in range(K):
        idx = (typ == k).nonzero().squeeze(1)
        x = h.double()[src[idx]]
        y = dagg.double()[dst[idx]]
        if weighted:
            y = y * w.double()[idx].unsqueeze(1)
        ref.append(y.t() @ x)
        val = x @ A.double()[k].t()
        if weighted:
            val = val * w.double()[idx].unsqueeze(1)
        fwd = fwd.index_add(0, dst[idx], val)
    ref = torch.stack(ref)
    assert max_err(agg.detach(), fwd) < 1e-5
    assert max_err(Al.grad, ref) / max(1.0, float(ref.abs().max())) < 2e-5
    # with gradients wanted for the node features too (the generic backward: d(msg), dx, transposed scatter)
    hl = h.clone().requires_grad_(True)
    agg2 = ops.message_aggregate(hl, A, g, w)
    agg2.backward(dagg)
    dh = torch.zeros(V, F, dtype=torch.float64, device=dev)
    for k in range(K):
        idx = (typ == k).nonzero().squeeze(1)
        y = dagg.double()[dst[idx]]
        if weighted:
            y = y * w.double()[idx].unsqueeze(1)
        dh = dh.index_add(0, src[idx], y @ A.double()[k])
    assert max_err(hl.grad, dh) / max(1.0, float(dh.abs().max())) < 1e-5


def test_c2_basic_model_training_step_against_float64(dev, c2):
    """The whole timed step on the c2 batch -- 3 rounds of message -> sum -> GRU with random (non-zero-bias) weights,
    backward from a random cotangent -- against a float64 restatement on the device: final node state and the gradients
    of every hot-path parameter."""
    from mpnn_amd.models.basic_model import BasicModel
    mb, g, h = c2
    V, H, T = g.num_nodes, 64, 3
    torch.manual_seed(5)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                       message_steps=T).to(dev)
    with torch.no_grad():
        # a tower that carries signal through its 50 aliased layers (kaiming, as test_lipo.py:132 initialises it),
        # non-zero biases everywhere (so A0 != 0), and a last layer scaled to put the bond matrices at O(1/4)
        for mod in model.mf.edge_map.modules():
            if isinstance(mod, torch.nn.Linear):
                torch.nn.init.kaiming_uniform_(mod.weight, nonlinearity="relu")
        for n, p in model.named_parameters():
            if n.endswith("bias") or "bias_" in n:
                p.uniform_(-0.05, 0.05)
        A_now, _ = model.mf._edge_matrices(g)
        last = model.mf.edge_map[-1]
        sc = 0.25 / float(A_now.abs().max())
        last.weight.mul_(sc)
        last.bias.mul_(sc)
    mask = torch.ones(V, 1, device=dev)
    cot = torch.randn(V, H, device=dev) / V ** 0.5
    state, _ = model.message_passing(h, g, g, mask)
    state.backward(cot)
    # float64 restatement: tower on the K+1 bond rows, per-type dense products, index_add neighbour sum, GRU
    p64 = {n: p.detach().double().requires_grad_(True) for n, p in model.named_parameters() if not n.startswith("of.")}
    rows = torch.cat([g.type_feat.new_zeros(1, 4), g.type_feat]).double()
    x = rows
    mods = list(model.mf.edge_map)
    i = 0
    while i < len(mods):
        mod = mods[i]
        if isinstance(mod, torch.nn.Linear):
            x = x @ p64["mf.edge_map.%d.weight" % i].t() + p64["mf.edge_map.%d.bias" % i]
        elif isinstance(mod, torch.nn.Sequential):
            first = next(j for j, mm in enumerate(mods) if mm is mod)        # the 50 aliases share one tensor
            x = torch.relu(x @ p64["mf.edge_map.%d.0.weight" % first].t())
        else:
            x = torch.relu(x)
        i += 1
    A = x.view(-1, H, H)[1:]
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    h64 = h.double()
    agg = torch.zeros(V, H, dtype=torch.float64, device=dev)
    for k in range(A.shape[0]):
        idx = (typ == k).nonzero().squeeze(1)
        agg = agg.index_add(0, dst[idx], h64[src[idx]] @ A[k].t())
    st = h64
    for _ in range(T):
        st = _gru_ref64(agg, st, mask.double(), p64["uf.gru_cell.weight_ih"], p64["uf.gru_cell.weight_hh"],
                        p64["uf.gru_cell.bias_ih"], p64["uf.gru_cell.bias_hh"])
    st.backward(cot.double())
    e_state = max_err(state.detach(), st.detach())
    record_parity("c2_full_size_training_step", node_state_max_abs_err=e_state, node_state_max_abs=float(st.detach().abs().max()),
                  atoms=V, hidden=H, steps=T, bar=1e-5)
    assert e_state < 1e-5                                     # measured 1.1e-6 (profiles/r04_parity.json)
    seen = set()
    for n, p in model.named_parameters():
        if n.startswith("of.") or p.data_ptr() in seen or n == "mf.message_bias":
            continue
        seen.add(p.data_ptr())
        first = n
        want = p64[first].grad
        if want is None:
            continue
        # aliased tower layers: the float64 graph accumulated every alias into the first name
        scale = max(1e-3, float(want.abs().max()))
        assert max_err(p.grad, want) / scale < 5e-4, n
