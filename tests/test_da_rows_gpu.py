"""The weight gradient of message + neighbour sum contracted over ATOMS (mpnn_message_agg_bwd_da_rows_f32,
csrc/edge_da_rows.hip; reference: the autograd of edge_network.py:40,52 composed with adjacent_message_agg.py:18) against
float64 and against the per-edge kernel: widths 64 / 128, one to four bond types, molecule batches and random graphs with
atoms of degree 0 ... 9, row counts that are no multiple of 32, gradients and features spread over twelve decades."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    if os.environ.get("MPNN_GRU_MATH") == "fp32":
        pytest.skip("a split-precision kernel")
    return torch.device("cuda:0")


def _ref64(h, dout, g, K):
    src, dst, typ = g.col_idx.long(), g.edge_dst.long(), g.edge_type.long()
    dA = torch.zeros(K, h.shape[1], h.shape[1], dtype=torch.float64, device=h.device)
    for k in range(K):
        sel = typ == k
        dA[k] = dout.double()[dst[sel]].t() @ h.double()[src[sel]]
    return dA


def _both(h, A, dout, g):
    from mpnn_amd import ops
    res = []
    for per_edge in (False, True):
        Ag = A.clone().requires_grad_(True)
        ops.DA_PER_EDGE = per_edge
        try:
            assert ops.da_rows_applies(Ag, None, None, g) == (not per_edge)
            out = ops.message_aggregate(h, Ag, g)
            (out * dout).sum().backward()
        finally:
            ops.DA_PER_EDGE = False
        res.append(Ag.grad)
    return res


@pytest.mark.parametrize("H", [64, 128])
@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("mols", [1, 333])
def test_per_atom_weight_gradient_on_molecules(dev, H, K, mols):
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    mb = synth.make_molecules(mols, H, seed=70 + K + mols, edge_features=K)
    g = MolGraph.from_molbatch(mb, dev)
    gen = torch.Generator(device=dev).manual_seed(H + K)
    V = g.num_nodes
    h = torch.randn(V, H, device=dev, generator=gen)
    A = torch.randn(K, H, H, device=dev, generator=gen) / H ** 0.5
    dout = torch.randn(V, H, device=dev, generator=gen)
    new, old = _both(h, A, dout, g)
    ref = _ref64(h, dout, g, K)
    scale = ref.abs().max()
    assert float((new.double() - ref).abs().max() / scale) < 2e-6
    assert float((old.double() - ref).abs().max() / scale) < 2e-6


@pytest.mark.parametrize("H,V", [(64, 2500), (128, 1111), (128, 31), (64, 33)])
@pytest.mark.parametrize("spread", [0.0, 6.0])
def test_per_atom_weight_gradient_on_random_graphs(dev, H, V, spread):
    """Degrees 0 ... 9 (the kernel requests three edges ahead and fetches the rest on the spot), sources anywhere, and -- with
    `spread` -- rows of the incoming gradient and of the features spread over 10^+-spread: every row block is split behind
    its own scales and the accumulators follow the running minimum."""
    from mpnn_amd.graph import MolGraph
    K = 4
    rng = np.random.default_rng(H + V + int(spread))
    deg = rng.integers(0, 10, V)
    row_ptr = np.zeros(V + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    E = int(row_ptr[-1])
    t = lambda a: torch.from_numpy(a).to(dev)
    g = MolGraph(t(row_ptr), t(rng.integers(0, V, E).astype(np.int32)), None, t(rng.integers(0, K, E).astype(np.int32)),
                 torch.zeros(K, 1, device=dev), torch.tensor([0, V], dtype=torch.int32, device=dev))
    gen = torch.Generator(device=dev).manual_seed(V)
    h = torch.randn(V, H, device=dev, generator=gen)
    dout = torch.randn(V, H, device=dev, generator=gen)
    if spread:
        h = h * torch.pow(10.0, (torch.rand(V, 1, device=dev, generator=gen) * 2 - 1) * spread)
        dout = dout * torch.pow(10.0, (torch.rand(V, 1, device=dev, generator=gen) * 2 - 1) * spread)
    A = torch.randn(K, H, H, device=dev, generator=gen) / H ** 0.5
    new, old = _both(h, A, dout, g)
    ref = _ref64(h, dout, g, K)
    for k in range(K):                                        # per matrix: the types' magnitudes differ with `spread`
        scale = ref[k].abs().max().clamp_min(1e-300)
        assert float((new[k].double() - ref[k]).abs().max() / scale) < 5e-6, (k, float((old[k].double() - ref[k]).abs().max() / scale))


def test_per_atom_weight_gradient_full_size(dev):
    """c4's shape: 125 k molecules at width 128; the two kernels agree and the new one matches float64 on a sample of entries."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MolGraph
    H, K = 128, 4
    mb = synth.make_molecules(125_000, H, seed=317, atom_features=False)
    g = MolGraph.from_molbatch(mb, dev)
    V = g.num_nodes
    h = synth.hashed_features(torch.arange(V, device=dev), H)
    gen = torch.Generator(device=dev).manual_seed(1)
    dout = torch.randn(V, H, device=dev, generator=gen) * 1e-3
    A = torch.randn(K, H, H, device=dev, generator=gen) / H ** 0.5
    new, old = _both(h, A, dout, g)
    ref = _ref64(h, dout, g, K)
    scale = ref.abs().max()
    e_new, e_old = (float((x.double() - ref).abs().max() / scale) for x in (new, old))
    from conftest import record_parity
    record_parity("c4_weight_gradient_of_message_sum", per_atom_kernel_max_abs_err_over_max=e_new,
                  per_edge_kernel_max_abs_err_over_max=e_old, atoms=float(V), bar=1e-5)
    assert e_new < 1e-5 and e_old < 1e-5                     # 3.7 M terms per entry, fp32 accumulators
