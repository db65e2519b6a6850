"""GRU forward at hidden 64 (c2 size; or 128 / 256 as argument): timing and float64 error of the running math mode.
    python tools/bench_gru_fwd.py [64|128|256]                   (default: two row-guarded fp16 pieces, three MFMAs per product)
    MPNN_GRU_MATH=fp32 python tools/bench_gru_fwd.py [64|128]    (strict fp32 MFMA)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops            # noqa: E402

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V = {64: 2_997_659, 128: 3_749_258, 256: 2_400_011}[H]
gen = torch.Generator(device=dev).manual_seed(1)
m = torch.randn(V, H, device=dev, generator=gen)
h = torch.rand(V, H, device=dev, generator=gen) * 2 - 1
mask = (torch.rand(V, 1, device=dev, generator=gen) > 0.1).float()
bound = (6.0 / (4 * H)) ** 0.5
W_ih = (torch.rand(H, 3 * H, device=dev, generator=gen) * 2 - 1) * bound
W_hh = (torch.rand(H, 3 * H, device=dev, generator=gen) * 2 - 1) * bound
b_ih = torch.rand(3 * H, device=dev, generator=gen) * 0.2 - 0.1
b_hh = torch.rand(3 * H, device=dev, generator=gen) * 0.2 - 0.1


def ref64(m, h, mask):
    gi = m.double() @ W_ih.double() + b_ih.double()
    gh = h.double() @ W_hh.double() + b_hh.double()
    mk = mask.double()
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mk
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mk
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mk
    return ((1 - z) * n + z * h.double()) * mk


for save in (False, True):
    fn = lambda: ops.gru_update_raw(m, h, mask.reshape(-1), W_ih, W_hh, b_ih, b_hh, save)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(15):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%s: best %.3f ms" % ("training (gates saved)" if save else "inference", best))
out, _ = ops.gru_update_raw(m, h, mask.reshape(-1), W_ih, W_hh, b_ih, b_hh, False)
n = 400_000
err = float((out[:n].double() - ref64(m[:n], h[:n], mask[:n])).abs().max())
print("max |out - float64| over %d atoms = %.3e" % (n, err))
# operands far from 1: m rows of 1e3, h rows of 1e-3 in one tile
m2, h2 = m[:n] * 1e3, h[:n] * 1e-3
out2, _ = ops.gru_update_raw(m2.contiguous(), h2.contiguous(), mask[:n].reshape(-1).contiguous(), W_ih, W_hh, b_ih, b_hh, False)
print("scaled operands: max err %.3e" % float((out2.double() - ref64(m2, h2, mask[:n])).abs().max()))
