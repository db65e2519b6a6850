"""Model-level parity at the SHAPES of BASELINE.json's configs (hidden 64 / 128 / 128-attention / 256,
their step counts, one-hot bond types), on batches small enough for the dense CPU oracle: forward,
final node state, and gradients of every parameter through message + aggregate + update.

The oracle itself is pinned by the reference-generated fixtures (tests/test_oracle_golden.py); these tests
extend the comparison to widths the fixtures do not carry (they would be tens of MB)."""
import numpy as np
import pytest
import torch

from conftest import max_err
from oracle import dense_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _small_batch(H, mols, seed, dist="drug", max_atoms=None):
    from mpnn_amd import synth
    mb = synth.make_molecules(200, H, seed=seed, dist=dist)
    ids = np.argsort(mb.n_atoms, kind="stable")[:mols] if max_atoms is None else \
        np.nonzero(mb.n_atoms <= max_atoms)[0][:mols]
    sub = synth.select(mb, ids)
    return {k: torch.from_numpy(v) for k, v in synth.to_dense(sub).items()}


def _rel(a, b):
    return max_err(a, b) / max(1.0, float(torch.as_tensor(b).detach().abs().max()))


def _shared_leaves(model):
    """CPU copies of a module's parameters as autograd leaves, one per distinct tensor, keyed like state_dict."""
    leaves, out = {}, {}
    for k, v in model.state_dict(keep_vars=True).items():
        if id(v) not in leaves:
            leaves[id(v)] = v.detach().cpu().clone().requires_grad_(v.requires_grad and v.is_floating_point())
        out[k] = leaves[id(v)]
    return out


@pytest.mark.parametrize("name,H,T,mols", [("c2", 64, 3, 6), ("c4", 128, 3, 4), ("c5", 256, 3, 2)])
def test_basic_model_at_config_width(dev, name, H, T, mols):
    from mpnn_amd.models.basic_model import BasicModel
    from mpnn_amd.models.graph_model_wrapper import GraphWrapper
    torch.manual_seed(317)
    batch = _small_batch(H, mols, seed=317 + H, dist="skewed" if name == "c5" else "drug", max_atoms=30 if name == "c5" else None)
    N = batch["adj"].shape[-1]
    model = GraphWrapper(BasicModel(H, 4, H, N, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={},
                                    message_steps=T))
    # biases away from zero so edge_map(0) != 0 takes part
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "bias" in k:
                p.uniform_(-0.1, 0.1)
    params = _shared_leaves(model)
    cot = torch.rand(mols, 8) - 0.5
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


def test_attention_model_at_c3_shape(dev):
    """configs[2]: att_model (AttEdgeNetwork per step + AdjMsgAgg + GRU + MaskBatchNorm), hidden 128, 5 steps."""
    from mpnn_amd.models.att_model import BasicModel as AttModel
    H, T, mols = 128, 5, 4
    torch.manual_seed(317)
    batch = _small_batch(H, mols, seed=99)
    N = batch["adj"].shape[-1]
    from mpnn_amd.mpnn_functions import GraphLevelOutput
    model = AttModel(H, 4, H, N, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T,
                     readout_func=GraphLevelOutput)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "bias" in k:
                p.uniform_(-0.1, 0.1)
    params = _shared_leaves(model)
    cot = torch.rand(mols, 8) - 0.5
    ref, ref_state = O.att_model_forward(params, batch["afm"], batch["bfm"], batch["adj"], batch["mask"], T, True)
    (ref * cot).sum().backward()
    model = model.to(dev)
    gb = {k: v.to(dev) for k, v in batch.items()}
    out = model(gb["afm"], gb["bfm"], gb["adj"], gb["mask"])
    state, _ = model.message_passing(gb["afm"], gb["bfm"], gb["adj"], gb["mask"])
    (out * cot.to(dev)).sum().backward()
    assert _rel(state.detach().cpu(), ref_state) < 2e-5
    assert _rel(out.detach().cpu(), ref) < 5e-5
    checked = 0
    for k, p in model.named_parameters():
        g_ref = params[k].grad
        if g_ref is None or p.grad is None:
            continue
        assert _rel(p.grad.cpu(), g_ref) < 5e-4, (k, _rel(p.grad.cpu(), g_ref))
        checked += 1
    assert checked >= 20


def test_hoisted_message_matches_per_step_message(dev):
    """BasicModel.hoist_message computes message + aggregate once per pass (their input is the constant afm,
    models/basic_model.py:57): the node states are bit-identical, the gradients equal up to summation order."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel
    H = 64
    mb = synth.make_molecules(400, H, seed=5)
    g = MolGraph.from_molbatch(mb, dev)
    torch.manual_seed(1)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=3).to(dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    mask = torch.ones(afm.shape[0], 1, device=dev)
    res = []
    for hoist in (False, True):
        model.hoist_message = hoist
        for p in model.parameters():
            p.grad = None
        state, _ = model.message_passing(afm, g, g, mask)
        state.square().sum().backward()
        res.append((state.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        scale = max(1.0, float(res[0][1][k].abs().max()))
        assert max_err(res[0][1][k], res[1][1][k]) / scale < 1e-5, k


@pytest.mark.parametrize("H", [64, 128, 256])
@pytest.mark.parametrize("case", ["single_atoms", "one_molecule", "isolated_atoms", "tiny_many"])
def test_degenerate_batches_at_fast_path_widths(dev, H, case):
    """Degenerate dense batches through the width-specific kernels (64 / 128 / 256): molecules of one atom (no edges
    at all), a single molecule, atoms without bonds inside a molecule, many two-atom molecules -- forward and the GRU /
    message parameter gradients against the oracle."""
    from mpnn_amd.models.basic_model import BasicModel
    g = torch.Generator().manual_seed(H + sum(map(ord, case)))
    if case == "single_atoms":
        B, N = 5, 1
        adj = torch.zeros(B, N, N)
    elif case == "one_molecule":
        B, N = 1, 9
        adj = (torch.rand(B, N, N, generator=g) < 0.3).float()
    elif case == "isolated_atoms":
        B, N = 3, 7
        adj = (torch.rand(B, N, N, generator=g) < 0.3).float()
        adj[:, 2, :] = 0
        adj[:, :, 2] = 0                                   # atom 2 of every molecule has no bond
    else:
        B, N = 40, 2
        adj = torch.ones(B, N, N)
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (1 - torch.eye(N))
    mask = torch.ones(B, N, 1)
    if case == "isolated_atoms":
        mask[1, 5:] = 0                                    # and molecule 1 is shorter (padded rows)
        adj[1, 5:, :] = 0
        adj[1, :, 5:] = 0
    types = torch.randint(0, 4, (B, N, N), generator=g)
    types = torch.triu(types, 1)
    types = types + types.transpose(1, 2)
    bfm = torch.nn.functional.one_hot(types, 4).float() * adj.unsqueeze(-1)
    afm = (torch.rand(B, N, H, generator=g) * 2 - 1) * mask
    torch.manual_seed(5)
    model = BasicModel(H, 4, H, N, 4, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=2)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "bias" in k:
                p.uniform_(-0.1, 0.1)
    params = _shared_leaves(model)
    cot = torch.rand(B, 4, generator=g) - 0.5
    ref, ref_state = O.basic_model_forward(params, afm, bfm, adj, mask, 2, True)
    (ref * cot).sum().backward()
    model = model.to(dev)
    out = model(afm.to(dev), bfm.to(dev), adj.to(dev), mask.to(dev))
    state, _ = model.message_passing(afm.to(dev), bfm.to(dev), adj.to(dev), mask.to(dev))
    (out * cot.to(dev)).sum().backward()
    assert _rel(state.detach().cpu(), ref_state) < 2e-5
    assert _rel(out.detach().cpu(), ref) < 5e-5
    for k, p in model.named_parameters():
        if params[k].grad is None or p.grad is None:
            continue
        assert _rel(p.grad.cpu(), params[k].grad) < 5e-4, (k, _rel(p.grad.cpu(), params[k].grad))
