"""Masked batch norms sitting between message and update in the lipo / attention models.

Reference: models/mask_batch_norm.py:5-38.  Quirks kept on purpose (parity):
* MaskBatchNorm1d divides by (sqrt(var) + eps) -- eps OUTSIDE the root -- uses the biased masked
  variance for the running estimate, and in eval mode divides by (running_var**0.5 + eps);
* MaskBatchNorm (no parameters) takes the mean of the UNMASKED sum over the masked count and
  normalises by sqrt(var + eps), eps = 1e-6.
Statistics are global reductions over all atoms of the batch (torch reductions; a fused two-pass
HIP kernel is a "next" row of the scope table).
"""
import torch
from torch import nn


def _flatten(tensor, mask):
    return tensor.reshape(-1, tensor.shape[-1]), mask.reshape(-1, 1)


class MaskBatchNorm(nn.Module):
    def forward(self, tensor, mask, eps=1e-6):
        y, mk = _flatten(tensor, mask)
        count = mk.sum()
        centred = (y - y.sum(dim=0) / count) * mk
        var = centred.pow(2).sum(dim=0) / count
        return (centred / torch.sqrt(var + eps)).view(tensor.shape)


class MaskBatchNorm1d(nn.BatchNorm1d):
    def forward(self, tensor, mask):
        y, mk = _flatten(tensor, mask)
        use_running = (not self.training) and self.track_running_stats
        if use_running:
            y = (y - self.running_mean) / (self.running_var ** .5 + self.eps)
        else:
            count = mk.sum()
            mean = (y * mk).sum(dim=0) / count
            var = ((y - mean) * mk).pow(2).sum(dim=0) / count
            if self.track_running_stats:
                with torch.no_grad():
                    keep = 1 - self.momentum
                    self.running_mean = keep * self.running_mean + self.momentum * mean
                    self.running_var = keep * self.running_var + self.momentum * var
            y = (y - mean) / (var.sqrt() + self.eps)
        if self.affine:
            y = self.weight * y + self.bias
        return (y * mk).view(tensor.shape)
