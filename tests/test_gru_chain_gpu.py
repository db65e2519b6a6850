"""BasicModel's T updates as one autograd node (ops.GRUChain; basic_model.py:57-59): same outputs as T separate update nodes
(the same kernels), same gradients up to the order in which the weight gradients are summed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("H,mols,T", [(22, 16, 3), (64, 300, 3), (128, 100, 4)])
def test_chained_updates_equal_separate_updates(dev, H, mols, T):
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    from mpnn_amd.models.basic_model import BasicModel
    mb = synth.make_molecules(mols, H, seed=11 + H, edge_features=4)
    g = MolGraph.from_molbatch(mb, dev)
    afm = torch.from_numpy(mb.atom_feat).to(dev)
    mask = torch.ones(afm.shape[0], 1, device=dev)
    torch.manual_seed(5)
    model = BasicModel(H, 4, H, 50, 8, message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=T).to(dev)
    res = []
    for chain in (True, False):
        model.chain_updates = chain
        model.zero_grad(set_to_none=True)
        out = model(afm, g, g, mask)
        out.square().sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0])                       # the same forward kernels on the same inputs
    assert res[0][1].keys() == res[1][1].keys()
    for k, gr in res[0][1].items():
        ref = res[1][1][k]
        assert float((gr - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) < 1e-5, k


def test_chain_on_a_dense_padded_batch(dev):
    import numpy as np
    from mpnn_amd import synth
    from mpnn_amd.models.basic_model import BasicModel
    H = 22
    mb = synth.make_molecules(6, H, seed=3, dist="lipo", edge_features=4)
    d = synth.to_dense(synth.select(mb, np.arange(6)))
    dense = {k: torch.from_numpy(v).to(dev) for k, v in d.items() if k in ("afm", "bfm", "adj", "mask")}
    torch.manual_seed(2)
    model = BasicModel(H, dense["bfm"].shape[-1], H, dense["adj"].shape[-1] if dense["adj"].dim() == 4 else 1, 3,
                       message_opts={}, agg_opts={}, update_opts={}, readout_opts={}, message_steps=3).to(dev)
    outs = []
    for chain in (True, False):
        model.chain_updates = chain
        outs.append(model(dense["afm"], dense["bfm"], dense["adj"], dense["mask"]).detach().clone())
    assert torch.equal(outs[0], outs[1])
