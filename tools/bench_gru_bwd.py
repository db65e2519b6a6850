#!/usr/bin/env python3
"""Time the GRU forward/backward kernels alone on c2-sized arrays (env knobs select variants)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops
dev = torch.device("cuda:0")
V, H = 2_997_659, 64
g = torch.Generator(device=dev).manual_seed(0)
m, h, dout = (torch.randn(V, H, device=dev, generator=g) for _ in range(3))
mask = torch.ones(V, device=dev)
W1, W2 = (torch.randn(H, 3 * H, device=dev, generator=g) / 8 for _ in range(2))
b1, b2 = (torch.randn(3 * H, device=dev, generator=g) / 8 for _ in range(2))
out, saved = ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("env", {k: v for k, v in os.environ.items() if k.startswith("MPNN_")})
print("fwd (save)   %.3f ms" % t(lambda: ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)))
print("fwd (nosave) %.3f ms" % t(lambda: ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, False)))
print("bwd          %.3f ms" % t(lambda: ops.gru_update_bwd_raw(dout, m, h, mask, W1, W2, saved)))
