#!/usr/bin/env python3
"""GRU forward/backward kernels alone at hidden 64 (c2 size), 128 (c4 size) or 256: timing and float64 error.
    python tools/bench_gru_bwd.py [128]                      (default: backward on two fp16 pieces, three MFMAs per product)
    MPNN_GRU_MATH=fp32 python tools/bench_gru_bwd.py [128]   (strict fp32 MFMA)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import ops
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V = {64: 2_997_659, 128: 3_749_258, 256: 2_400_011}[H]
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
if len(sys.argv) > 2 and sys.argv[2] == "time":          # timing only (under rocprofv3)
    sys.exit(0)


def ref64(m, h, mask, dout):
    m64, h64 = m.double().requires_grad_(True), h.double().requires_grad_(True)
    W1d, W2d, b1d, b2d = (x.double().requires_grad_(True) for x in (W1, W2, b1, b2))
    gi, gh = m64 @ W1d + b1d, h64 @ W2d + b2d
    mk = mask.double().reshape(-1, 1)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]) * mk
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H]) * mk
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) * mk
    o = ((1 - z) * n + z * h64) * mk
    o.backward(dout.double())
    return m64.grad, h64.grad, W1d.grad, W2d.grad, b1d.grad, b2d.grad


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))


# float64 error over the first n atoms, three gradient profiles: unit scale; tiny gradients; magnitudes that climb / fall
# by 1e8 across the rows (the per-tile range guards of the fp16 variant change along the way)
n = 300_001
mk = (torch.rand(n, device=dev, generator=g) > 0.1).float()
ramp = torch.logspace(-8, 0, n, device=dev).reshape(-1, 1)
for name, d in (("unit", dout[:n]), ("x1e-6", dout[:n] * 1e-6), ("rising 1e-8..1", dout[:n] * ramp),
                ("falling 1..1e-8", dout[:n] * ramp.flip(0))):
    d = d.contiguous()
    o, sv = ops.gru_update_raw(m[:n].contiguous(), h[:n].contiguous(), mk, W1, W2, b1, b2, True)
    got = ops.gru_update_bwd_raw(d, m[:n].contiguous(), h[:n].contiguous(), mk, W1, W2, sv)
    want = ref64(m[:n], h[:n], mk, d)
    print("%-16s max err / max |ref|: dm %.2e dh %.2e dW_ih %.2e dW_hh %.2e db_ih %.2e db_hh %.2e" %
          ((name,) + tuple(rel(a, b) for a, b in zip(got, want))))
    # per-row accuracy of dm, dh: the worst row's error relative to that row's largest entry
    for k, nm in ((0, "dm"), (1, "dh")):
        e = (got[k].double() - want[k]).abs().amax(1)
        s = want[k].abs().amax(1)
        live = s > 0
        print("    %s worst row: %.2e" % (nm, float((e[live] / s[live]).max())))
