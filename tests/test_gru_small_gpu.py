"""The vector-pipe GRU kernels of the small widths (csrc/gru_small.hip: H <= 40, the Lipophilicity model's 22-38 features;
reference: mpnn_functions/update/gru_update.py:13-35 and its autograd) against the oracle: every width class, atom counts
from 1 up, partial masks, the reference driver's batch size (~430 atoms) and a large batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("V,H", [(1, 22), (5, 1), (9, 7), (430, 22), (431, 33), (777, 38), (1000, 40), (64, 32), (30011, 22)])
def test_small_width_gru_forward_and_backward(dev, V, H):
    from oracle import dense_ref as O
    from mpnn_amd import ops
    g = torch.Generator().manual_seed(V * 5 + H)
    mk = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)
    m, h, cot = mk(V, H), mk(V, H), mk(V, H)
    mask = (torch.rand(V, generator=g) < 0.8).float()
    s = 1.0 / np.sqrt(H)
    names = ("gru_cell.weight_ih", "gru_cell.weight_hh", "gru_cell.bias_ih", "gru_cell.bias_hh")
    p = {names[0]: mk(H, 3 * H) * s, names[1]: mk(H, 3 * H) * s, names[2]: mk(3 * H) * 0.5, names[3]: mk(3 * H) * 0.5}
    leaves = [t.double().requires_grad_(True) for t in (m, h, *[p[n] for n in names])]
    ref = O.gru_update(dict(zip(names, leaves[2:])), leaves[0], leaves[1], mask.double().view(-1, 1))
    gref = torch.autograd.grad((ref * cot.double()).sum(), leaves)
    dl = [t.clone().to(dev).requires_grad_(True) for t in (m, h, *[p[n] for n in names])]
    out = ops.gru_update(dl[0], dl[1], mask.to(dev), *dl[2:])
    ggpu = torch.autograd.grad((out * cot.to(dev)).sum(), dl)
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-6
    for a, b, name in zip(ggpu, gref, ("dm", "dh") + names):
        assert _rel(a.cpu(), b) < (5e-6 if V < 10000 else 2e-5), (name, _rel(a.cpu(), b))   # (atomically accumulated dW at 30 k atoms)


def test_small_width_kernels_are_the_ones_that_run(dev):
    """A launch at width 22 takes microseconds, not the 17-20 (forward) / ~70 (backward) of the matrix-pipe kernels padded to
    their instruction shapes: 200 forward + backward pairs at the reference driver's batch size take ~12 ms, host side
    included (a loose bound: the box's host speed varies)."""
    from mpnn_amd import ops
    V, H = 430, 22
    g = torch.Generator(device=dev).manual_seed(0)
    m, h, dout = (torch.randn(V, H, device=dev, generator=g) for _ in range(3))
    mask = torch.ones(V, device=dev)
    W1, W2 = (torch.randn(H, 3 * H, device=dev, generator=g) / 5 for _ in range(2))
    b1, b2 = (torch.randn(3 * H, device=dev, generator=g) / 5 for _ in range(2))
    out, saved = ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)

    def pair():
        ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)
        ops.gru_update_bwd_raw(dout, m, h, mask, W1, W2, saved)
    for _ in range(20):
        pair()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        pair()
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) < 60.0, e0.elapsed_time(e1)
