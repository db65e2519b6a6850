#!/usr/bin/env python3
"""Reads the in-kernel cycle stamps of a -DRC_STAMP build of csrc/gru_bwd_rc.hip (diagnostic only, its own library directory):
    MPNN_EXTRA_HIPCC_FLAGS=-DRC_STAMP python -m mpnn_amd.build && MPNN_EXTRA_HIPCC_FLAGS=-DRC_STAMP python tools/stamp_gru_rc.py
Average cycles per step of block 3: the dm | dh kernel's consumer wave 0 and producer wave 4; per tile of the dW kernel's
waves 0 and 7 (block 16)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpnn_amd import _lib, ops  # noqa: E402

H = 128
V = 3_749_258
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
m, h, dout = (torch.randn(V, H, device=dev, generator=g) for _ in range(3))
mask = torch.ones(V, device=dev)
W1, W2 = (torch.randn(H, 3 * H, device=dev, generator=g) / 8 for _ in range(2))
b1, b2 = (torch.randn(3 * H, device=dev, generator=g) / 8 for _ in range(2))
out, saved = ops.gru_update_raw(m, h, mask, W1, W2, b1, b2, True)
lib = ctypes.CDLL(_lib.load()._name)
for _ in range(3):
    ops.gru_update_bwd_raw(dout, m, h, mask, W1, W2, saved)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib.mpnn_debug_rc_stamps(buf, 1)
ops.gru_update_bwd_raw(dout, m, h, mask, W1, W2, saved)
torch.cuda.synchronize()
lib.mpnn_debug_rc_stamps(buf, 0)
n = max(buf[7], 1)
print("dm|dh consumer 0, %d steps, cycles/step: note reads %.0f | rescale+direct term %.0f | 72 MFMAs %.0f | copy wait %.0f | epilogue %.0f | B1 %.0f | copy issue + B2 %.0f | sum %.0f"
      % ((buf[7],) + tuple(buf[i] / n for i in range(7)) + (sum(buf[i] for i in range(7)) / n,)))
n = max(buf[21], 1)
print("dm|dh producer 4, %d steps, cycles/step: gate math %.0f | row requests %.0f | B1 %.0f | park %.0f | B2 %.0f | sum %.0f"
      % ((buf[21],) + tuple(buf[16 + i] / n for i in range(5)) + (sum(buf[16 + i] for i in range(5)) / n,)))
for wv, b in ((0, 32), (7, 44)):
    n = max(buf[b + 7], 1)
    print("dW wave %d, %d tiles, cycles/tile: barrier %.0f | gate pieces %.0f | row requests+rescale %.0f | 36 MFMAs %.0f | publish %.0f | barrier %.0f | park x %.0f | sum %.0f"
          % ((wv, buf[b + 7]) + tuple(buf[b + i] / n for i in range(7)) + (sum(buf[b + i] for i in range(7)) / n,)))
